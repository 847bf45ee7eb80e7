// dqp_al.hip -- Newton step of the augmented-Lagrangian MPC solver.
//
// Replaces, per Newton step of qpth/al_utils.py NewtonAL.forward (al_utils.py:403-427):
//   merit_hess = diag(Q) + rho * Jc^T Jc            (al_utils.py:96-102,176-178: a bmm per step)
//   U, info    = torch.linalg.cholesky_ex(merit_hess)
//   update     = -torch.cholesky_solve(grad, U)
// and, in NewtonAL.backward (al_utils.py:465-482),  inp_grad = -cholesky_solve(x_grad, U).
//
// One problem per 256-thread workgroup (4 wavefronts); the nz x nz Hessian lives in LDS
// (nz <= 128 -> <= 132 KB).  Jc^T Jc is the one real GEMM on this path (M = N = nz, K = ncon,
// e.g. 100 x 100 x 120 for cartpole T=20), so it runs on the fp64 matrix cores:
// v_mfma_f64_16x16x4_f64 tiles, lower-triangular tile pairs dealt round-robin to the 4 waves;
// a lane's A and B operands are both plain coalesced reads of one row segment of Jc.
// Jc is staged through LDS once (coalesced HBM reads); the factorisation (square-root-free,
// H = M D M^T) keeps the matrix 16-cyclic in the registers of a 16x16 thread grid and only
// passes the pivot column through LDS, one barrier per step; the two triangular solves run on
// one wavefront with the vector in registers; L = M sqrt(D) is written out (zero upper part,
// like torch) for backward.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/dqp.h"
#include "dqp_r16_prims.h"
#include "dqp_dyn_models.h"   // row_sum (DPP row reduction)

#ifdef DQP_STAMPS
namespace dqp { extern unsigned long long *g_debug_stamps; }
#endif

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

struct AlP {
    const double *Jc, *Qd, *rho, *grad, *Lin, *rhs;
    double *update, *L, *out;
    int32_t *info;
    int B, nz, ncon, ld, chunk_rows, flag_off;
    unsigned long long *stamps;   // diagnostic (only read with -DDQP_STAMPS; tools/stamps_al.py)
};

#ifdef DQP_STAMPS
#define AL_STAMP(i) do { if (P.stamps && threadIdx.x == 0) P.stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AL_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ double bcast64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rcp_nr(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    return fma(fma(-d, r, 1.0), r, r);
}

// x <- (M D M^T)^-1 b  (UNIT: LDS holds the unit-lower M below the diagonal and sqrt(d) on it) or
// x <- (L L^T)^-1 b    (!UNIT: LDS holds the Cholesky factor), on ONE wavefront, no barriers.
// Element i of the vector lives on lane i & 63, slot i >> 6 (nz <= 128).  Columns are fetched
// eight steps ahead of the substitution chain, so a step costs one readlane pair and one FMA.
template <bool UNIT, bool FORWARD = true>
__device__ __forceinline__ void wave_solve(const double *H, int ld, int nz, int zidx, double (&b)[2], int lane)
{
    double dg[2], rdg[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int i = lane + 64 * s;
        dg[s] = i < nz ? H[i * ld + i] : 1.0;
        rdg[s] = 1.0 / dg[s];
    }
    // branch-free: a masked entry reads the zero word the caller keeps at H[zidx]
    constexpr int G = 8;
    double mc[G][2], mn[G][2];
    auto load_fwd = [&](double (&m)[G][2], int k0) {     // column k = k0 + u, rows i > k
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int k = k0 + u;
            m[u][0] = H[(lane > k && lane < nz) ? lane * ld + k : zidx];
            m[u][1] = H[(lane + 64 > k && lane + 64 < nz) ? (lane + 64) * ld + k : zidx];
        }
    };
    auto load_bwd = [&](double (&m)[G][2], int k0) {     // row k = k0 - u, columns i < k
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int k = k0 - u;
            m[u][0] = H[(lane < k && k < nz) ? k * ld + lane : zidx];
            m[u][1] = H[(lane + 64 < k && k < nz) ? k * ld + lane + 64 : zidx];
        }
    };
    // eight chained substitution steps on columns/rows k0 + sgn*u, pivots in slot S (k >> 6 == S)
#define DQP_AL_CHAIN(S, SGN)                                                              \
    _Pragma("unroll") for (int u = 0; u < G; ++u) {                                       \
        const int k = k0 + (SGN) * u;                                                     \
        double wk = bcast64(b[S], k & 63);                                                \
        if (!UNIT) {                                                                      \
            wk *= bcast64(rdg[S], k & 63);                                                \
            b[S] = (lane + 64 * (S) == k) ? wk : b[S];                                    \
        }                                                                                 \
        b[0] = fma(-mc[u][0], wk, b[0]);                                                  \
        b[1] = fma(-mc[u][1], wk, b[1]);                                                  \
    }
#define DQP_AL_ROTATE() _Pragma("unroll") for (int u = 0; u < G; ++u) { mc[u][0] = mn[u][0]; mc[u][1] = mn[u][1]; }
    const int nzg = ((nz + G - 1) / G) * G, lo_end = min(nzg, 64);
    if (FORWARD) {
        load_fwd(mn, 0);
        for (int k0 = 0; k0 < lo_end; k0 += G) { DQP_AL_ROTATE(); load_fwd(mn, k0 + G); DQP_AL_CHAIN(0, 1); }
        for (int k0 = 64; k0 < nzg; k0 += G) { DQP_AL_ROTATE(); load_fwd(mn, k0 + G); DQP_AL_CHAIN(1, 1); }
    }
    if (UNIT) { b[0] *= rdg[0] * rdg[0]; b[1] *= rdg[1] * rdg[1]; }   // D^-1, d = (sqrt d)^2
    load_bwd(mn, nzg - 1);
    for (int k0 = nzg - 1; k0 >= 64; k0 -= G) { DQP_AL_ROTATE(); load_bwd(mn, k0 - G); DQP_AL_CHAIN(1, -1); }
    for (int k0 = lo_end - 1; k0 >= 0; k0 -= G) { DQP_AL_ROTATE(); load_bwd(mn, k0 - G); DQP_AL_CHAIN(0, -1); }
#undef DQP_AL_CHAIN
#undef DQP_AL_ROTATE
}

#define TRI(a, b) ((a) * ((a) + 1) / 2 + (b))

// lower-triangular 16x16 tiles in row-major order: tile t -> (ti, tj), tj <= ti
__host__ __device__ constexpr int tile_row(int t) { int ti = 0; while ((ti + 1) * (ti + 2) / 2 <= t) ++ti; return ti; }
__host__ __device__ constexpr int tile_col(int t) { return t - tile_row(t) * (tile_row(t) + 1) / 2; }

// K-loop over one staged chunk for wave W: tiles W, W+4, ... accumulate in registers; the NT
// column-block operands of a 4-row K step are fetched one step ahead of the MFMAs using them.
template <int NT, int W, int NS>
__device__ __forceinline__ void mfma_chunk(const double *Hs, int rows4, int kq, int l15, double4_t (&acc)[NS])
{
    constexpr int NZP = 16 * NT, NTILES = NT * (NT + 1) / 2;
    const double *jr = Hs + kq * NZP + l15;
    double opc[NT], opn[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) opn[c] = jr[16 * c];
    for (int c0 = 0; c0 < rows4; c0 += 4) {
#pragma unroll
        for (int c = 0; c < NT; ++c) opc[c] = opn[c];
        const double *jn = jr + (c0 + 4 < rows4 ? c0 + 4 : c0) * NZP;
#pragma unroll
        for (int c = 0; c < NT; ++c) opn[c] = jn[16 * c];
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (W + 4 * s < NTILES)
                acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(opc[tile_row(W + 4 * s)], opc[tile_col(W + 4 * s)],
                                                              acc[s], 0, 0, 0);
    }
}

// C/D layout of the f64 16x16x4 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
template <int NT, int W, int NS>
__device__ __forceinline__ void store_tiles(double *H, int ld, int nz, double rho, int kq, int l15,
                                            const double4_t (&acc)[NS])
{
    constexpr int NTILES = NT * (NT + 1) / 2;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (W + 4 * s >= NTILES) continue;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = 16 * tile_row(W + 4 * s) + kq + 4 * rg, col = 16 * tile_col(W + 4 * s) + l15;
            if (row < nz && col <= row) H[row * ld + col] = rho * acc[s][rg];
        }
    }
}

// One problem per 256-thread workgroup; NT = ceil(nz / 16) is a template parameter so every
// register-array loop (MFMA tiles, the 16-cyclic factor) is fully unrolled at its true size.
template <int NT>
__global__ __launch_bounds__(256) void al_newton_kernel(AlP P)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int NZP = 16 * NT, NS = (NT * (NT + 1) / 2 + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ty = tid >> 4, tx = tid & 15;
    const int nz = P.nz, ncon = P.ncon, ld = P.ld;
    const long long prob = blockIdx.x;
    double *H = sm;                                   // nz x ld
    if (tid == 0) sm[P.flag_off + 1] = 0.0;           // the zero word masked solve loads read
    AL_STAMP(0);
    // diagonal of the cost Hessian for the 16-cyclic owner of (i, i): threads with ty == tx
    double qd[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) qd[a] = (ty == tx && ty + 16 * a < nz) ? P.Qd[prob * nz + ty + 16 * a] : 0.0;

    // ---- H = diag(Qd) + rho * Jc^T Jc ---------------------------------------------------------
    // Jc is staged through LDS (the not-yet-written H region) in zero-padded chunks of CH rows,
    // read from HBM exactly once with coalesced loads; each wave owns up to 9 lower-triangular
    // 16x16 output tiles whose fp64 MFMA accumulators stay in registers across the whole K loop.
    const double *J = P.Jc + prob * (long long)ncon * nz;
    const double rho = P.rho[prob];
    const int CH = P.chunk_rows;
    double4_t acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int l15 = lane & 15, kq = lane >> 4;
    for (int cb = 0; cb < ncon; cb += CH) {
        const int rows = min(CH, ncon - cb), rows4 = (rows + 3) & ~3;
        __syncthreads();
        // wave w stages rows w, w+4, ...; a lane covers columns lane and lane+64; 16 loads in flight
        for (int r0 = wave; r0 < rows4; r0 += 32) {
            double v[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int rr = r0 + 4 * u, cc = lane + 64 * s;
                    v[u][s] = (rr < rows && cc < nz) ? J[(long long)(cb + rr) * nz + cc] : 0.0;
                }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int rr = r0 + 4 * u, cc = lane + 64 * s;
                    if (rr < rows4 && cc < NZP) H[rr * NZP + cc] = v[u][s];
                }
        }
        __syncthreads();
        switch (wave) {
        case 0: mfma_chunk<NT, 0, NS>(H, rows4, kq, l15, acc); break;
        case 1: mfma_chunk<NT, 1, NS>(H, rows4, kq, l15, acc); break;
        case 2: mfma_chunk<NT, 2, NS>(H, rows4, kq, l15, acc); break;
        default: mfma_chunk<NT, 3, NS>(H, rows4, kq, l15, acc); break;
        }
    }
    __syncthreads();
    AL_STAMP(1);
    switch (wave) {
    case 0: store_tiles<NT, 0, NS>(H, ld, nz, rho, kq, l15, acc); break;
    case 1: store_tiles<NT, 1, NS>(H, ld, nz, rho, kq, l15, acc); break;
    case 2: store_tiles<NT, 2, NS>(H, ld, nz, rho, kq, l15, acc); break;
    default: store_tiles<NT, 3, NS>(H, ld, nz, rho, kq, l15, acc); break;
    }
    __syncthreads();
    AL_STAMP(2);

    // ---- factorisation in registers --------------------------------------------------------------
    // Thread (ty, tx) of the 16 x 16 grid owns the 16-cyclic elements H[ty + 16a][tx + 16b],
    // b <= a.  Step k: every thread picks up the pivot and its <= NT row and <= NT column
    // operands from the published column k (double-buffered in LDS, one barrier per step) and
    // applies the rank-1 update to its registers -- next pivot column first, so its owners
    // publish it before the bulk of the update (look-ahead).  Columns stay unscaled (c_ik) with
    // the pivot d_k on the diagonal: H = M D M^T, M_ik = c_ik / d_k.
    const bool fused = nz < 16 * NT;       // row nz exists in the 16-cyclic grid (nz not a multiple of 16)
    double hr[NT * (NT + 1) / 2];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            const int row = ty + 16 * a, col = tx + 16 * b;
            hr[TRI(a, b)] = (row < nz && col <= row) ? H[row * ld + col] : 0.0;
            if (a == b) hr[TRI(a, b)] += qd[a];
            // augmented row nz = rhs^T: the elimination then carries out the forward substitution
            // M w = -grad for free (row nz ends up holding w), so wave 0 only has the backward sweep
            if (fused && row == nz && col < nz) hr[TRI(a, b)] = -P.grad[prob * nz + col];
        }
    __syncthreads();
    double *colbuf = sm;                 // 2 x 128 doubles, aliases the (now unused) H region
    double dcol[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) dcol[b] = 1.0;
    int firstbad = 0;                    // torch.linalg.cholesky_ex: info = first non-positive minor
    if (tx == 0) {
#pragma unroll
        for (int a = 0; a < NT; ++a) colbuf[ty + 16 * a] = hr[TRI(a, 0)];
    }
#pragma unroll
    for (int kb = 0; kb < NT; ++kb) {
        const int kend = min(16, nz - 16 * kb);
        for (int kr = 0; kr < kend; ++kr) {
            const int k = 16 * kb + kr;
            const double *cbuf = colbuf + (k & 1) * 128;
            double *nbuf = colbuf + ((k + 1) & 1) * 128;
            __syncthreads();
            const double d = cbuf[k];
            double r[NT], c[NT];
#pragma unroll
            for (int a = kb; a < NT; ++a) { r[a] = cbuf[ty + 16 * a]; c[a] = cbuf[tx + 16 * a]; }
            firstbad = (firstbad == 0 && !(d > 0.0)) ? k + 1 : firstbad;
            const double nrd = -rcp_nr(d);
            if (tx == kr) dcol[kb] = d;
            r[kb] = ty > kr ? r[kb] : 0.0;
            c[kb] = tx > kr ? c[kb] : 0.0;
#pragma unroll
            for (int a = kb; a < NT; ++a) r[a] *= nrd;
#pragma unroll
            for (int a = kb; a < NT; ++a) hr[TRI(a, kb)] = fma(r[a], c[kb], hr[TRI(a, kb)]);
            if (kr < 15) {
                if (tx == kr + 1) {
#pragma unroll
                    for (int a = kb; a < NT; ++a) nbuf[ty + 16 * a] = hr[TRI(a, kb)];
                }
            }
            if (kb + 1 < NT) {
#pragma unroll
                for (int a = kb + 1; a < NT; ++a) hr[TRI(a, kb + 1)] = fma(r[a], c[kb + 1], hr[TRI(a, kb + 1)]);
                if (kr == 15) {
                    if (tx == 0) {
#pragma unroll
                        for (int a = kb + 1; a < NT; ++a) nbuf[ty + 16 * a] = hr[TRI(a, kb + 1)];
                    }
                }
            }
#pragma unroll
            for (int b = kb + 2; b < NT; ++b)
#pragma unroll
                for (int a = b; a < NT; ++a) hr[TRI(a, b)] = fma(r[a], c[b], hr[TRI(a, b)]);
        }
    }
    __syncthreads();
    AL_STAMP(3);
    // LDS <- unit-lower M (strictly below the diagonal) and sqrt(d) on the diagonal
    {
        double rdc[NT], sdc[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) { rdc[b] = 1.0 / dcol[b]; sdc[b] = sqrt(dcol[b]); }
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                const int row = ty + 16 * a, col = tx + 16 * b;
                if (row < nz && col <= row) H[row * ld + col] = col == row ? sdc[b] : hr[TRI(a, b)] * rdc[b];
            }
    }
    double *wvec = sm + P.flag_off + 2;    // row nz of the elimination = w (fused forward sweep)
    if (fused && ty == (nz & 15)) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
            if ((nz >> 4) >= b && tx + 16 * b < nz) {
#pragma unroll
                for (int a = 0; a < NT; ++a)
                    if (a >= b && a == (nz >> 4)) wvec[tx + 16 * b] = hr[TRI(a, b)];
            }
    }
    __syncthreads();
    const int info = firstbad;
    AL_STAMP(4);

    // ---- wave 0 solves while waves 1-3 write the factor out -------------------------------------
    if (wave == 0) {
        double b[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int i = lane + 64 * s;
            b[s] = i < nz ? (fused ? wvec[i] : -P.grad[prob * nz + i]) : 0.0;
        }
        if (info == 0) {
            if (fused) wave_solve<true, false>(H, ld, nz, P.flag_off + 1, b, lane);
            else wave_solve<true, true>(H, ld, nz, P.flag_off + 1, b, lane);
        }
        const double nanv = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (lane + 64 * s < nz) P.update[prob * nz + lane + 64 * s] = info == 0 ? b[s] : nanv;
        if (lane == 0 && P.info) P.info[prob] = info;
        AL_STAMP(5);
    } else if (P.L) {                     // Cholesky factor L_ik = M_ik sqrt(d_k), zero upper part
        double *Lo = P.L + prob * (long long)nz * nz;
        double sd[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) sd[s] = H[min(lane + 64 * s, nz - 1) * (ld + 1)];
        for (int i = wave - 1; i < nz; i += 3)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int j = lane + 64 * s;
                if (j < nz) Lo[i * nz + j] = j < i ? H[i * ld + j] * sd[s] : (j == i ? sd[s] : 0.0);
            }
    }
}

// out = -(L L^T)^-1 rhs   (NewtonAL.backward, al_utils.py:477-480); block loads L, wave 0 solves
__global__ __launch_bounds__(256) void al_chol_solve_kernel(AlP P)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nz = P.nz, ld = P.ld;
    const long long prob = blockIdx.x;
    double *H = sm;
    const double *Li = P.Lin + prob * (long long)nz * nz;
    if (tid == 0) sm[P.flag_off + 1] = 0.0;
    for (int i = wave; i < nz; i += 4)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int j = lane + 64 * s;
            if (j <= i) H[i * ld + j] = Li[i * nz + j];
        }
    __syncthreads();
    if (tid >= 64) return;
    double b[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) b[s] = (lane + 64 * s < nz) ? -P.rhs[prob * nz + lane + 64 * s] : 0.0;
    wave_solve<false>(H, ld, nz, P.flag_off + 1, b, lane);
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (lane + 64 * s < nz) P.out[prob * nz + lane + 64 * s] = b[s];
}

int fill(const dqp_al_dims *d, AlP &P, size_t &lds)
{
    if (!d || d->nbatch < 0 || d->nz <= 0 || d->ncon < 0) return DQP_ERR_BAD_ARG;
    if (d->nz > 128) return DQP_ERR_TOO_LARGE;
    P.B = d->nbatch; P.nz = d->nz; P.ncon = d->ncon;
#ifdef DQP_STAMPS
    P.stamps = dqp::g_debug_stamps;
#else
    P.stamps = nullptr;
#endif
    // LDS: the nz x nz factor (ld = nz | 1: 80,800 B at nz = 100, so two problems share a CU); the
    // same region first stages Jc in chunks of `chunk_rows` zero-padded rows of 16*ceil(nz/16).
    P.ld = d->nz | 1;                 // odd leading dimension: conflict-free column reads
    const int nzp = (d->nz + 15) & ~15;
    size_t hd = (size_t)d->nz * P.ld;
    if (hd < (size_t)4 * nzp) hd = (size_t)4 * nzp;
    if (hd < 256) hd = 256;           // the double-buffered pivot column of the factorisation
    P.chunk_rows = (int)((hd / nzp) & ~(size_t)3);
    P.flag_off = (int)hd;
    lds = (hd + 2 + 128) * sizeof(double);   // + zero word, + the fused forward-sweep result
    if (lds > 160 * 1024 - 64) return DQP_ERR_TOO_LARGE;
    return DQP_OK;
}

// ------------------------------------------------------------------------------------------
// Constraint Jacobian fill + J^T (lam + rho res_c)  (al_utils.py:62-102,162-318)
struct AsmP {
    const double *Jx, *Ju, *lam, *resc, *rho;
    double *Jc, *gterm;
    int B, n, m, T;
    const double *Qd, *q, *xu;       // optional: gterm += Qd * xu + q  (the full merit gradient)
};

// value of J[row][col] for one problem; act: inequality rows count only when active
__device__ __forceinline__ double jac_entry(const AsmP &P, const double *Jx, const double *Ju,
                                            const double *resc, int row, int col, bool clamp)
{
    const int n = P.n, m = P.m, T = P.T, nt = n + m, neq = T * n;
    const int tc = col / nt, jc = col - tc * nt;
    if (row < neq) {
        const int t = row / n, r = row - t * n;
        if (t == T - 1) return (tc == 0 && jc == r) ? 1.0 : 0.0;                 // x_0 - x0
        if (tc == t) return jc < n ? -Jx[(t * n + r) * n + jc] : -Ju[(t * n + r) * m + (jc - n)];
        return (tc == t + 1 && jc == r) ? 1.0 : 0.0;                            // + x_{t+1}
    }
    const int q = row - neq, t = q / (2 * m), k = q - t * 2 * m;                 // [upper (m), lower (m)]
    if (clamp && !(resc[row] > 0.0)) return 0.0;
    if (tc != t || jc < n) return 0.0;
    const int i = jc - n;
    return k < m ? (i == k ? 1.0 : 0.0) : (i == k - m ? -1.0 : 0.0);
}

__global__ __launch_bounds__(256) void al_assemble_kernel(AsmP P)
{
    const int n = P.n, m = P.m, T = P.T, nt = n + m, nz = T * nt, neq = T * n, ncon = neq + 2 * T * m;
    const long long b = blockIdx.x;
    const double *Jx = P.Jx + b * (long long)(T - 1) * n * n, *Ju = P.Ju + b * (long long)(T - 1) * n * m;
    const double *resc = P.resc + b * (long long)ncon;
    if (P.Jc) {
        double *Jc = P.Jc + b * (long long)ncon * nz;
        for (int e = threadIdx.x; e < ncon * nz; e += 256) {
            const int row = e / nz, col = e - row * nz;
            Jc[e] = jac_entry(P, Jx, Ju, resc, row, col, true);
        }
    }
    if (P.gterm) {
        // column sums with mu = lam + rho res_c (for an inactive inequality res_c = 0, so the
        // clamped and the full Jacobian give the same term): only the structurally non-zero rows
        const double *lam = P.lam + b * (long long)ncon;
        const double rho = P.rho[b];
        for (int col = threadIdx.x; col < nz; col += 256) {
            const int t = col / nt, j = col - t * nt;
            auto mu = [&](int row) { return lam[row] + rho * resc[row]; };
            double acc = 0.0;
            if (t < T - 1)
                for (int r = 0; r < n; ++r) acc += jac_entry(P, Jx, Ju, resc, t * n + r, col, false) * mu(t * n + r);
            if (j < n) {
                if (t >= 1) acc += mu((t - 1) * n + j);
                if (t == 0) acc += mu((T - 1) * n + j);
            } else {
                const int i = j - n;
                acc += mu(neq + t * 2 * m + i) - mu(neq + t * 2 * m + m + i);
            }
            if (P.Qd) acc += P.Qd[b * (long long)nz + col] * P.xu[b * (long long)nz + col] + P.q[b * (long long)nz + col];
            P.gterm[b * (long long)nz + col] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------
// merit function of a batch of candidate trajectories (al_utils.py:37-59); 16 lanes per
// (candidate, problem): lane j of the group sweeps knots j, j+16, ... and the partial sums are
// combined with DPP row reductions
struct MeritP {
    const double *xu, *xn, *x0, *Qd, *q, *lam, *rho, *ul, *uu;
    double *merit;
    int B, n, m, T, ncand;
};

__global__ __launch_bounds__(256) void al_merit_kernel(MeritP P)
{
    const int n = P.n, m = P.m, T = P.T, nt = n + m, neq = T * n, ncon = neq + 2 * T * m;
    const long long item = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);      // (cand, problem)
    const int r = threadIdx.x & 15;
    const long long total = (long long)P.ncand * P.B;
    const long long it = item < total ? item : total - 1;
    const long long b = it % P.B;
    const double *xu = P.xu + it * (long long)T * nt, *xn = P.xn + it * (long long)(T - 1) * n;
    const double *Qd = P.Qd + b * (long long)T * nt, *q = P.q + b * (long long)T * nt;
    const double *lam = P.lam + b * (long long)ncon, *x0 = P.x0 + b * (long long)n;
    const double rho = P.rho[b];
    double acc = 0.0;
    for (int t = r; t < T; t += 16) {
        const double *z = xu + t * nt;
        for (int j = 0; j < nt; ++j) acc += (0.5 * Qd[t * nt + j] * z[j] + q[t * nt + j]) * z[j];
        for (int j = 0; j < n; ++j) {                       // equality row block of knot t
            double res;
            int row;
            if (t < T - 1) { res = xu[(t + 1) * nt + j] - xn[t * n + j]; row = t * n + j; }
            else { res = xu[j] - x0[j]; row = (T - 1) * n + j; }
            acc += (0.5 * rho * res + lam[row]) * res;
        }
        for (int i = 0; i < m; ++i) {                       // box rows of knot t
            const double u = z[n + i], up = u - P.uu[i], lo = P.ul[i] - u;
            const int row = neq + t * 2 * m + i;
            acc += lam[row] * up + lam[row + m] * lo + 0.5 * rho * (fmax(up, 0.0) * fmax(up, 0.0) + fmax(lo, 0.0) * fmax(lo, 0.0));
        }
    }
    acc = dqp::r16::row_sum(acc);
    if (r == 0 && item < total) P.merit[item] = acc;
}


// ------------------------------------------------------------------------------------------
// NewtonAL with a registered device model: the Python glue of al_utils.NewtonAL.forward
// (qpth/al_utils.py:363-456, 503-527) as kernels, so that the four Newton steps of one AL iteration
// are 21 back-to-back launches with no host involvement.
template <class Map>
__device__ __forceinline__ void lin_knot(const double *z, double dt, double *xn, double *Jx, double *Ju)
{
    constexpr int NX = Map::NX, NU = Map::NU, K = NX + NU;
    using S = dqp::dyn::Dual<K>;
    S xa[NX], ua[NU], o[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) { xa[k] = S(z[k]); xa[k].d[k] = 1.0; }
#pragma unroll
    for (int k = 0; k < NU; ++k) { ua[k] = S(z[NX + k]); ua[k].d[NX + k] = 1.0; }
    Map::template step<S>(xa, ua, dt, o);
#pragma unroll
    for (int r = 0; r < NX; ++r) {
        xn[r] = o[r].v;
#pragma unroll
        for (int c = 0; c < NX; ++c) Jx[r * NX + c] = o[r].d[c];
#pragma unroll
        for (int c = 0; c < NU; ++c) Ju[r * NU + c] = o[r].d[NX + c];
    }
}
template <class Map>
__device__ __forceinline__ void step_knot(const double *xs, const double *us, double dt, double *xn)
{
    double xa[Map::NX], ua[Map::NU], o[Map::NX];
#pragma unroll
    for (int k = 0; k < Map::NX; ++k) xa[k] = xs[k];
#pragma unroll
    for (int k = 0; k < Map::NU; ++k) ua[k] = us[k];
    Map::template step<double>(xa, ua, dt, o);
#pragma unroll
    for (int k = 0; k < Map::NX; ++k) xn[k] = o[k];
}

struct LinP {
    const double *xu, *x0, *ul, *uu;
    double *Jx, *Ju, *resc;
    double dt;
    int B, n, m, T, dyn;
};

// one thread per (problem, knot): dynamics + Jacobians of the knot, its equality rows
// x_{t+1} - f(x_t, u_t) (or x_0 - x0 for the last block) and its clamped box rows
__global__ __launch_bounds__(256) void al_linearize_kernel(LinP P)
{
    const int n = P.n, m = P.m, T = P.T, nt = n + m, neq = T * n, ncon = neq + 2 * T * m;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)P.B * T) return;
    const long long b = idx / T;
    const int t = (int)(idx - b * T);
    const double *z = P.xu + (b * T + t) * nt;
    double *resc = P.resc + b * ncon;
    if (t < T - 1) {
        double xn[8], Jx[64], Ju[16];
        switch (P.dyn) {
        case DQP_DYN_PENDULUM1L: lin_knot<dqp::dyn::Robot<dqp::dyn::Pendulum1l>>(z, P.dt, xn, Jx, Ju); break;
        case DQP_DYN_CARTPOLE1L: lin_knot<dqp::dyn::Robot<dqp::dyn::Cartpole1l>>(z, P.dt, xn, Jx, Ju); break;
        case DQP_DYN_CARTPOLE2L: lin_knot<dqp::dyn::Robot<dqp::dyn::Cartpole2l>>(z, P.dt, xn, Jx, Ju); break;
        case DQP_DYN_PENDULUM_EULER: lin_knot<dqp::dyn::PendulumEuler>(z, P.dt, xn, Jx, Ju); break;
        default: lin_knot<dqp::dyn::PendulumDx>(z, P.dt, xn, Jx, Ju); break;
        }
        double *oJx = P.Jx + (b * (T - 1) + t) * n * n, *oJu = P.Ju + (b * (T - 1) + t) * n * m;
        for (int i = 0; i < n * n; ++i) oJx[i] = Jx[i];
        for (int i = 0; i < n * m; ++i) oJu[i] = Ju[i];
        for (int i = 0; i < n; ++i) resc[t * n + i] = z[nt + i] - xn[i];
    } else {
        const double *z0 = P.xu + b * T * nt;
        for (int i = 0; i < n; ++i) resc[(T - 1) * n + i] = z0[i] - P.x0[b * n + i];
    }
    for (int i = 0; i < m; ++i) {
        const double u = z[n + i];
        resc[neq + t * 2 * m + i] = fmax(u - P.uu[i], 0.0);
        resc[neq + t * 2 * m + m + i] = fmax(P.ul[i] - u, 0.0);
    }
}

struct LsAP {
    const double *xu, *upd, *x0, *Qd, *q, *lam, *rho, *ul, *uu;
    double *merit;
    double dt;
    int B, n, m, T, ncand, dyn;
    // optional: selection folded into the group kernel (argmin, acceptance, update of the iterate;
    // al_utils.py:516-526) when one group sees all candidates of its problem
    double *xu_w, *merit_cur, *status;
    int32_t *fail;
    const int32_t *info;
};

// merit (al_utils.py:37-59) of ncand candidates xu + 2^-k upd per problem (x_0 pinned to x0,
// al_utils.py:515), dynamics evaluated in the kernel; 16 lanes per (candidate, problem).  ncand = 0
// evaluates the current point (step 0) into merit[b].  Knots are processed 16 at a time:
//   phase A  element-wise over the entries of knots c .. c+16, lanes along the contiguous axis
//            (coalesced): z = xu + step upd -> LDS; cost and box-constraint terms of knots c .. c+15
//   phase B  one knot per lane: the model step from the knot in LDS (odd stride: conflict-free),
//            residual against the next knot in LDS, multiplier and penalty terms
template <class Map>
__global__ __launch_bounds__(256) void al_ls_kernel(LsAP P)
{
    constexpr int n = Map::NX, m = Map::NU, nt = n + m, ZS = nt | 1;
    __shared__ double ls_lds[16 * 17 * ZS];
    const int T = P.T, neq = T * n, ncon = neq + 2 * T * m;
    const int nc = P.ncand > 0 ? P.ncand : 1;
    const int slot = threadIdx.x >> 4, r = threadIdx.x & 15;
    const long long item = (long long)blockIdx.x * 16 + slot;
    const long long total = (long long)nc * P.B;
    const long long it = item < total ? item : total - 1;
    const long long b = it % P.B;
    const int k = (int)(it / P.B);
    const double step = P.ncand > 0 ? (double)exp2f(-(float)k) : 0.0;      // float steps, as the reference
    const double *xu = P.xu + b * (long long)T * nt, *up = P.upd + b * (long long)T * nt;
    const double *Qd = P.Qd + b * (long long)T * nt, *q = P.q + b * (long long)T * nt;
    const double *lam = P.lam + b * (long long)ncon, *x0 = P.x0 + b * (long long)n;
    const double rho = P.rho[b];
    double *zb = ls_lds + slot * 17 * ZS;
    double acc = 0.0;
    for (int c = 0; c < T; c += 16) {
        const int kend = (c + 17 < T ? c + 17 : T) - c;                  // knots staged this round (<= 17)
#pragma unroll 4
        for (int e = r; e < kend * nt; e += 16) {
            const int tl = e / nt, j = e - tl * nt, t = c + tl, ge = c * nt + e;
            double z = xu[ge] + step * up[ge];
            if (t == 0 && j < n) z = x0[j];
            zb[tl * ZS + j] = z;
            if (tl < 16) {
                acc += (0.5 * Qd[ge] * z + q[ge]) * z;
                if (j >= n) {
                    const int i = j - n, row = neq + t * 2 * m + i;
                    const double hi = z - P.uu[i], lo = P.ul[i] - z;
                    acc += lam[row] * hi + lam[row + m] * lo +
                           0.5 * rho * (fmax(hi, 0.0) * fmax(hi, 0.0) + fmax(lo, 0.0) * fmax(lo, 0.0));
                }
            }
        }
        __syncthreads();
        const int t = c + r;
        if (t < T - 1) {
            double z[nt], xn[n];
#pragma unroll
            for (int j = 0; j < nt; ++j) z[j] = zb[r * ZS + j];
            Map::template step<double>(z, z + n, P.dt, xn);
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const double res = zb[(r + 1) * ZS + j] - xn[j];
                acc += (0.5 * rho * res + lam[t * n + j]) * res;
            }
        }
        __syncthreads();
    }
    // (the x_0 rows contribute nothing: x_0 is pinned to x0, their residual is exactly zero)
    acc = dqp::r16::row_sum(acc);
    if (r == 0 && item < total) P.merit[item] = acc;
}

// The same merits with one lane GROUP per problem that walks its candidates itself (T <= 32): the
// problem's xu, upd, Qd, q and multipliers are loaded once into registers instead of once per
// candidate, and the group is as wide as the horizon needs (TPI = 8, 16 or 32 lanes: one knot per lane in
// the model phase -- with 16 lanes per candidate a T = 5 problem used 5 of them).
// sum over a group of TPI lanes (aligned in the wavefront), in every lane of the group: DPP steps inside a 16-lane row
// (a shuffle is a ds_bpermute round trip through the LDS crossbar, ~100 cycles each on the chain of a candidate), one
// shuffle across the two rows of a 32-lane group
template <int TPI> __device__ __forceinline__ double group_sum(double v)
{
    using dqp::r16::dppd;
    if constexpr (TPI >= 16) {
        v = dqp::r16::row_sum(v);
#pragma unroll
        for (int off = 16; off < TPI; off <<= 1) v += __shfl_xor(v, off, 64);
    } else if constexpr (TPI == 8) {
        v += dppd<0x141>(v);        // row_half_mirror: l <-> 7 - l
        v += dppd<0xb1>(v);         // quad_perm [1,0,3,2]
        v += dppd<0x4e>(v);         // quad_perm [2,3,0,1]
    } else {
        static_assert(TPI == 4, "lane groups of 4, 8, 16 or 32");
        v += dppd<0xb1>(v);
        v += dppd<0x4e>(v);
    }
    return v;
}

// A lane group (TPI <= 32 lanes) never spans wavefronts and a wavefront's LDS operations execute in order:
// the group's writes to its own LDS rows only have to be issued before its reads, which a workgroup-scope fence
// (an s_waitcnt, no s_barrier) plus a scheduling barrier guarantees -- the four wavefronts of the block do not
// wait for each other twice per candidate.
__device__ __forceinline__ void group_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <class Map, int TPI, int TMAX>
__global__ __launch_bounds__(256, ((TMAX * (Map::NX + Map::NU) + TPI - 1) / TPI >= 12) ? 2 : 1) void al_ls_group_kernel(LsAP P)
{
    constexpr int n = Map::NX, m = Map::NU, nt = n + m, ZS = nt | 1;
    constexpr int EPL = (TMAX * nt + TPI - 1) / TPI;       // elements of the (T, nt) arrays per lane, T <= TMAX
    extern __shared__ double lsg_lds[];
    const int T = P.T, neq = T * n, ncon = neq + 2 * T * m, nzq = T * nt;
    const int nc = P.ncand > 0 ? P.ncand : 1;
    const int grp = threadIdx.x / TPI, r = threadIdx.x % TPI;
    const long long pb = (long long)blockIdx.x * (256 / TPI) + grp;
    const bool live = pb < P.B;
    const long long b = live ? pb : P.B - 1;
    const double *xu = P.xu + b * (long long)nzq, *up = P.upd + b * (long long)nzq;
    const double *Qd = P.Qd + b * (long long)nzq, *q = P.q + b * (long long)nzq;
    const double *lam = P.lam + b * (long long)ncon, *x0 = P.x0 + b * (long long)n;
    const double rho = P.rho[b];
    double *zb = lsg_lds + (size_t)grp * T * ZS;
    // ---- the problem, once: element e = r + TPI i of the (T, nt) arrays; the box rows (multipliers, bounds) only exist
    // for the T m control elements and are distributed over the lanes on their own (element c = r + TPI i of the (T, m)
    // array, with its own copy of the iterate and the update): eight arrays of EPL doubles per lane were 144 - 256
    // registers before the model's own, one wavefront per SIMD; four + six short ones leave room for two
    constexpr int UPL = (TMAX * m + TPI - 1) / TPI;
    // Long knots (EPL >= 12: the quadrotor's 16 elements per lane): the quadratic cost along the search direction is
    // the polynomial c0 + s c1 + s^2 c2 of the step -- three numbers per lane instead of the cost's two arrays (64
    // registers at nt = 16, which kept the kernel at one wavefront per SIMD) and two fmas per candidate instead of 2 EPL.
    constexpr bool POLY = EPL >= 12;
    double ex[EPL], eu[EPL], eq[POLY ? 1 : EPL], el[POLY ? 1 : EPL];
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int e = r + TPI * i, ec = e < nzq ? e : 0, t = ec / nt, j = ec - t * nt;
        const bool ok = e < nzq;
        ex[i] = ok ? xu[ec] : 0.0; eu[i] = (ok && P.ncand > 0) ? up[ec] : 0.0;
        const double qd = ok ? Qd[ec] : 0.0, ql = ok ? q[ec] : 0.0;
        if (ok && t == 0 && j < n) { ex[i] = x0[j]; eu[i] = 0.0; }            // x_0 pinned to x0 (al_utils.py:515)
        if constexpr (POLY) {
            const double qx = qd * ex[i];
            c0 += (0.5 * qx + ql) * ex[i];
            c1 += (qx + ql) * eu[i];
            c2 += 0.5 * qd * eu[i] * eu[i];
        } else {
            eq[i] = qd; el[i] = ql;
        }
    }
    double cx[UPL], cu[UPL], lu[UPL], ll[UPL], hi[UPL], lo[UPL];
#pragma unroll
    for (int i = 0; i < UPL; ++i) {
        const int c = r + TPI * i, cc = c < T * m ? c : 0, t = cc / m, iu = cc - t * m;
        const bool ok = c < T * m;
        const int ec = t * nt + n + iu, row = neq + t * 2 * m + iu;
        cx[i] = ok ? xu[ec] : 0.0; cu[i] = (ok && P.ncand > 0) ? up[ec] : 0.0;
        lu[i] = ok ? lam[row] : 0.0; ll[i] = ok ? lam[row + m] : 0.0;
        hi[i] = ok ? P.uu[iu] : INFINITY; lo[i] = ok ? P.ul[iu] : -INFINITY;
    }
    // multipliers of the lane's dynamics rows: registers, or -- long knots -- the group's LDS rows behind its iterate
    double ly[POLY ? 1 : n];
    double *lyb = lsg_lds + (size_t)(256 / TPI) * T * ZS + ((size_t)grp * T + r) * n;
#pragma unroll
    for (int j = 0; j < n; ++j) {
        const double v = (r < T - 1) ? lam[r * n + j] : 0.0;
        if constexpr (POLY) { if (r < T) lyb[j] = v; } else ly[j] = v;
    }
    // candidates k = blockIdx.y, blockIdx.y + gridDim.y, ...: small batches are split over more wavefronts
    const bool fold = P.xu_w != nullptr && gridDim.y == 1 && P.ncand > 0;
    double best = 0.0;
    int arg = 0;
    bool isnan_ = false;
    for (int k = blockIdx.y; k < nc; k += gridDim.y) {
        const double step = P.ncand > 0 ? (double)exp2f(-(float)k) : 0.0;    // float steps, as the reference
        double acc = POLY ? fma(step, fma(step, c2, c1), c0) : 0.0;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int e = r + TPI * i;
            if (e < nzq) {
                const int t = e / nt, j = e - t * nt;
                const double z = fma(step, eu[i], ex[i]);
                zb[t * ZS + j] = z;
                if constexpr (!POLY) acc += (0.5 * eq[i] * z + el[i]) * z;
            }
        }
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            if (r + TPI * i < T * m) {
                const double z = fma(step, cu[i], cx[i]);
                const double vh = z - hi[i], vl = lo[i] - z;
                acc += lu[i] * vh + ll[i] * vl + 0.5 * rho * (fmax(vh, 0.0) * fmax(vh, 0.0) + fmax(vl, 0.0) * fmax(vl, 0.0));
            }
        }
        group_sync();
        if (r < T - 1) {
            double z[nt], xn[n];
#pragma unroll
            for (int j = 0; j < nt; ++j) z[j] = zb[r * ZS + j];
            Map::template step<double>(z, z + n, P.dt, xn);
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const double res = zb[(r + 1) * ZS + j] - xn[j];
                acc += (0.5 * rho * res + (POLY ? lyb[j] : ly[j])) * res;
            }
        }
        group_sync();
        acc = group_sum<TPI>(acc);
        if (fold) {             // torch.min over the candidates: NaN wins, else the first minimum
            if (k == 0) { best = acc; arg = 0; isnan_ = acc != acc; }
            else if (!isnan_) {
                if (acc != acc) { best = acc; arg = k; isnan_ = true; }
                else if (acc < best) { best = acc; arg = k; }
            }
        } else if (r == 0 && live) {
            P.merit[(long long)k * P.B + b] = acc;
        }
    }
    if (fold) {
        const bool take = best < P.merit_cur[b];
        if (r == 0 && live) {
            P.merit_cur[b] = best;                              // new_merit regardless of acceptance
            if (P.status) P.status[b] = take ? 1.0 : 0.0;
            if (P.info && P.info[b] != 0) atomicOr(P.fail, 1);  // Cholesky failed: the caller re-runs the slow path
        }
        if (take && live) {
            const double sb = (double)exp2f(-(float)arg);
            double *xw = P.xu_w + b * (long long)nzq;
#pragma unroll
            for (int i = 0; i < EPL; ++i) {
                const int e = r + TPI * i;
                if (e < nzq) xw[e] = fma(sb, eu[i], ex[i]);
            }
        }
    }
}

template <class Map, int TPI, int TMAX = TPI> int launch_ls_group(const LsAP &P, hipStream_t st)
{
    constexpr int ZS = (Map::NX + Map::NU) | 1, G = 256 / TPI;
    constexpr bool POLY = (TMAX * (Map::NX + Map::NU) + TPI - 1) / TPI >= 12;
    const size_t lds = (size_t)G * P.T * (ZS + (POLY ? Map::NX : 0)) * sizeof(double);
    const unsigned blocks = (unsigned)((P.B + G - 1) / G);
    // at least ~2 wavefronts per SIMD where the batch alone does not give them: split the candidates
    unsigned split = 1;
    if (P.ncand > 1) while (split < 4 && (unsigned long long)blocks * 4 * split < 2048) split *= 2;
    DQP_LAUNCH((al_ls_group_kernel<Map, TPI, TMAX>), dim3(blocks, split), dim3(256), lds, st, P);
    return (P.xu_w && split == 1 && P.ncand > 0) ? 2 : DQP_OK;     // 2: the selection happened in the kernel
}

template <class Map> int launch_ls_t(const LsAP &P, hipStream_t st)
{
    // the model phase keeps T - 1 lanes of a group busy: four lanes for the shortest horizons (config 5: T = 5)
    if (P.T <= 5) return launch_ls_group<Map, 4, 5>(P, st);
    if (P.T <= 8) return launch_ls_group<Map, 8>(P, st);
    if (P.T <= 16) return launch_ls_group<Map, 16>(P, st);
    if (P.T <= 32) return launch_ls_group<Map, 32>(P, st);
    const long long items = (long long)(P.ncand > 0 ? P.ncand : 1) * P.B;
    DQP_LAUNCH(al_ls_kernel<Map>, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, st, P);
    return DQP_OK;
}

int launch_ls(const LsAP &P, hipStream_t st)
{
    switch (P.dyn) {
    case DQP_DYN_PENDULUM1L: return launch_ls_t<dqp::dyn::Robot<dqp::dyn::Pendulum1l>>(P, st);
    case DQP_DYN_CARTPOLE1L: return launch_ls_t<dqp::dyn::Robot<dqp::dyn::Cartpole1l>>(P, st);
    case DQP_DYN_CARTPOLE2L: return launch_ls_t<dqp::dyn::Robot<dqp::dyn::Cartpole2l>>(P, st);
    case DQP_DYN_PENDULUM_EULER: return launch_ls_t<dqp::dyn::PendulumEuler>(P, st);
    case DQP_DYN_PENDULUM_DX: return launch_ls_t<dqp::dyn::PendulumDx>(P, st);
    case DQP_DYN_REXQUADROTOR: return launch_ls_t<dqp::dyn::RexQuadrotor>(P, st);
    default: return DQP_ERR_BAD_ARG;
    }
}

struct OutP {
    const double *xu, *x0, *lam, *rho, *Qd, *q, *ul, *uu;
    double *lam_new, *cost, *resn;
    double dt;
    int B, n, m, T, dyn;
    double *rho_next;       // optional: rho * 10 (AL_mpc.py:307), for the next AL iteration of dqp_al_mpc_solve
};

// Between two AL iterations (qpth/AL_mpc.py:296-307): res = constraint residual at the new iterate,
// lam <- lam + rho res with the inequality block clamped at 0, cost of the iterate and the norm of
// the clamped residual; 16 lanes per problem, dynamics evaluated in the kernel.
template <class Map>
__global__ __launch_bounds__(256) void al_outer_kernel(OutP P)
{
    constexpr int n = Map::NX, m = Map::NU, nt = n + m;
    const int T = P.T, neq = T * n, ncon = neq + 2 * T * m;
    const long long item = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int r = threadIdx.x & 15;
    const long long b = item < P.B ? item : P.B - 1;
    const bool live = item < P.B;
    const double *xu = P.xu + b * (long long)T * nt, *lam = P.lam + b * (long long)ncon;
    const double *Qd = P.Qd + b * (long long)T * nt, *q = P.q + b * (long long)T * nt;
    double *ln = P.lam_new + b * (long long)ncon;
    const double rho = P.rho[b];
    double cost = 0.0, rn2 = 0.0;
    for (int t = r; t < T; t += 16) {
        double z[nt], xn[n];
#pragma unroll
        for (int j = 0; j < nt; ++j) z[j] = xu[t * nt + j];
        for (int j = 0; j < nt; ++j) cost += (0.5 * Qd[t * nt + j] * z[j] + q[t * nt + j]) * z[j];
        if (t < T - 1) {
            Map::template step<double>(z, z + n, P.dt, xn);
            for (int j = 0; j < n; ++j) {
                const double res = xu[(t + 1) * nt + j] - xn[j];
                rn2 += res * res;
                if (live) ln[t * n + j] = lam[t * n + j] + rho * res;
            }
        } else {
            for (int j = 0; j < n; ++j) {
                const double res = xu[j] - P.x0[b * n + j];
                rn2 += res * res;
                if (live) ln[(T - 1) * n + j] = lam[(T - 1) * n + j] + rho * res;
            }
        }
        for (int i = 0; i < m; ++i) {
            const double u = z[n + i], hi = u - P.uu[i], lo = P.ul[i] - u;
            const int row = neq + t * 2 * m + i;
            rn2 += fmax(hi, 0.0) * fmax(hi, 0.0) + fmax(lo, 0.0) * fmax(lo, 0.0);
            if (live) {
                ln[row] = fmax(lam[row] + rho * hi, 0.0);               // AL_mpc.py:300-301
                ln[row + m] = fmax(lam[row + m] + rho * lo, 0.0);
            }
        }
    }
    cost = dqp::r16::row_sum(cost);
    rn2 = dqp::r16::row_sum(rn2);
    if (live && r == 0) {
        P.cost[b] = cost; P.resn[b] = sqrt(rn2);
        if (P.rho_next) P.rho_next[b] = rho * 10.0;
    }
}

// Head of AL_mpc.MPC.al_solve (AL_mpc.py:254-283): xu = [x_init | u_init], cost_start = compute_cost(xu), the warm
// start of the multipliers / penalty from the previous call's history (al_utils.warm_start_al, al_utils.py:16-34:
// the newest stored AL iterate whose cost was already below cost_start, the newest one if none was; lam rescaled to
// that iterate's norm), and entry 0 of this call's history.  16 lanes per problem.
struct StartP {
    const double *x_init, *u_init, *Qd, *q, *lam_in, *rho_in;
    const double *hist_cost, *hist_lam, *hist_rho;      // previous call, oldest first: (K, B), (K, B, ncon), (K, B); or NULL
    double *xu, *cost0, *lam0, *rho0;
    int B, n, m, T, K;
    int32_t *fail;          // al_iter Cholesky-failure flags, cleared here
    int nfail;
};
__global__ __launch_bounds__(256) void al_start_kernel(StartP P)
{
    const int n = P.n, m = P.m, nt = n + m, T = P.T, ncon = T * n + 2 * T * m;
    const long long item = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int r = threadIdx.x & 15;
    const long long b = item < P.B ? item : P.B - 1;
    const bool live = item < P.B;
    if (blockIdx.x == 0 && (int)threadIdx.x < P.nfail) P.fail[threadIdx.x] = 0;
    double *xu = P.xu + b * (long long)T * nt;
    const double *Qd = P.Qd + b * (long long)T * nt, *q = P.q + b * (long long)T * nt;
    double quad = 0.0, lin = 0.0;
    for (int e = r; e < T * nt; e += 16) {
        const int t = e / nt, j = e - t * nt;
        const double v = j < n ? P.x_init[(b * T + t) * n + j] : P.u_init[(b * T + t) * m + (j - n)];
        if (live) xu[e] = v;
        quad += v * Qd[e] * v;
        lin += q[e] * v;
    }
    const double cost0 = 0.5 * dqp::r16::row_sum(quad) + dqp::r16::row_sum(lin);       // al_utils.compute_cost, diagonal cost
    const double *lam = P.lam_in + b * (long long)ncon;
    double scale = 1.0, rho = P.rho_in[b];
    if (P.K > 0) {
        int pick = P.K - 1;                                   // torch.max of an all-False column: index 0 = the newest
        for (int k = P.K - 1; k >= 0; --k)
            if (P.hist_cost[(long long)k * P.B + b] < cost0) { pick = k; break; }
        const double *lh = P.hist_lam + ((long long)pick * P.B + b) * ncon;
        double nh = 0.0, nl = 0.0;
        for (int e = r; e < ncon; e += 16) { nh += lh[e] * lh[e]; nl += lam[e] * lam[e]; }
        scale = sqrt(dqp::r16::row_sum(nh)) / sqrt(dqp::r16::row_sum(nl));
        rho = P.hist_rho[(long long)pick * P.B + b];
    }
    if (!live) return;
    double *l0 = P.lam0 + b * (long long)ncon;
    for (int e = r; e < ncon; e += 16) l0[e] = P.K > 0 ? lam[e] * scale : lam[e];
    if (r == 0) { P.cost0[b] = cost0; P.rho0[b] = rho; }
}

int launch_outer(const OutP &P, hipStream_t st)
{
    const dim3 grid((unsigned)((P.B + 15) / 16)), block(256);
    switch (P.dyn) {
    case DQP_DYN_PENDULUM1L: DQP_LAUNCH(al_outer_kernel<dqp::dyn::Robot<dqp::dyn::Pendulum1l>>, grid, block, 0, st, P); break;
    case DQP_DYN_CARTPOLE1L: DQP_LAUNCH(al_outer_kernel<dqp::dyn::Robot<dqp::dyn::Cartpole1l>>, grid, block, 0, st, P); break;
    case DQP_DYN_CARTPOLE2L: DQP_LAUNCH(al_outer_kernel<dqp::dyn::Robot<dqp::dyn::Cartpole2l>>, grid, block, 0, st, P); break;
    case DQP_DYN_PENDULUM_EULER: DQP_LAUNCH(al_outer_kernel<dqp::dyn::PendulumEuler>, grid, block, 0, st, P); break;
    case DQP_DYN_PENDULUM_DX: DQP_LAUNCH(al_outer_kernel<dqp::dyn::PendulumDx>, grid, block, 0, st, P); break;
    case DQP_DYN_REXQUADROTOR: DQP_LAUNCH(al_outer_kernel<dqp::dyn::RexQuadrotor>, grid, block, 0, st, P); break;
    default: return DQP_ERR_BAD_ARG;
    }
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

struct SelP {
    const double *merit, *upd, *x0;
    double *xu, *merit_cur, *status;
    int32_t *fail;
    const int32_t *info;
    int B, n, nz, ncand;
};

// argmin over the candidates, acceptance test and update of the iterate (al_utils.py:516-526):
// one 64-thread block per problem
__global__ __launch_bounds__(64) void al_select_kernel(SelP P)
{
    const long long b = blockIdx.x;
    // torch.min over the candidates (NaN wins, else the first minimum): lane k holds candidate k -- one round of
    // loads and a wavefront reduction instead of one lane walking ncand dependent loads (11 us -> launch-bound)
    const int lane = threadIdx.x;
    const bool has = lane < P.ncand;
    const double v = has ? P.merit[(long long)lane * P.B + b] : INFINITY;
    const unsigned long long nanmask = __builtin_amdgcn_ballot_w64(has && v != v);
    double m = (v != v) ? INFINITY : v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_xor(m, off, 64));
    const unsigned long long eq = __builtin_amdgcn_ballot_w64(has && v == m);
    const int arg = nanmask ? (int)__builtin_ctzll(nanmask) : (eq ? (int)__builtin_ctzll(eq) : 0);
    const double best = __shfl(v, arg, 64);
    const bool take = best < P.merit_cur[b];
    const double s_step = (double)exp2f(-(float)arg);
    if (lane == 0) {
        P.merit_cur[b] = best;                              // new_merit regardless of acceptance
        if (P.status) P.status[b] = take ? 1.0 : 0.0;
        if (P.info && P.info[b] != 0) atomicOr(P.fail, 1);  // Cholesky failed: the caller re-runs the slow path
    }
    if (!take) return;
    double *xu = P.xu + b * (long long)P.nz;
    const double *up = P.upd + b * (long long)P.nz;
    for (int e = threadIdx.x; e < P.nz; e += 64)
        xu[e] = e < P.n ? P.x0[b * P.n + e] : xu[e] + s_step * up[e];
}

template <typename K>
int launch(K kernel, const AlP &P, size_t lds, void *stream, int threads = 256)
{
    if (P.B == 0) return DQP_OK;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DQP_ERR_LAUNCH;
    DQP_LAUNCH(kernel, dim3(P.B), dim3(threads), lds, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int
dqp_al_newton_step(const dqp_al_dims *dims, const double *Jc, const double *Qdiag, const double *rho,
                   const double *grad, double *update, double *L, int32_t *info, void *stream)
{
    AlP P = {};
    size_t lds = 0;
    int rc = fill(dims, P, lds);
    if (rc) return rc;
    if (P.B == 0) return DQP_OK;
    if ((!Jc && P.ncon > 0) || !Qdiag || !rho || !grad || !update) return DQP_ERR_BAD_ARG;
    P.Jc = Jc; P.Qd = Qdiag; P.rho = rho; P.grad = grad; P.update = update; P.L = L; P.info = info;
    switch ((P.nz + 15) / 16) {
    case 1: return launch(al_newton_kernel<1>, P, lds, stream);
    case 2: return launch(al_newton_kernel<2>, P, lds, stream);
    case 3: return launch(al_newton_kernel<3>, P, lds, stream);
    case 4: return launch(al_newton_kernel<4>, P, lds, stream);
    case 5: return launch(al_newton_kernel<5>, P, lds, stream);
    case 6: return launch(al_newton_kernel<6>, P, lds, stream);
    case 7: return launch(al_newton_kernel<7>, P, lds, stream);
    default: return launch(al_newton_kernel<8>, P, lds, stream);
    }
}

__attribute__((visibility("default"))) int
dqp_al_chol_solve(const dqp_al_dims *dims, const double *L, const double *rhs, double *out, void *stream)
{
    AlP P = {};
    size_t lds = 0;
    int rc = fill(dims, P, lds);
    if (rc) return rc;
    if (P.B == 0) return DQP_OK;
    if (!L || !rhs || !out) return DQP_ERR_BAD_ARG;
    P.Lin = L; P.rhs = rhs; P.out = out;
    return launch(al_chol_solve_kernel, P, lds, stream);
}

__attribute__((visibility("default"))) int
dqp_al_assemble(const dqp_al_mpc_dims *d, const double *Jx, const double *Ju, const double *lam,
                const double *res_c, const double *rho, double *Jc, double *gterm, void *stream)
{
    if (!d || d->nbatch < 0 || d->n_state <= 0 || d->n_ctrl <= 0 || d->T < 2) return DQP_ERR_BAD_ARG;
    if (d->nbatch == 0) return DQP_OK;
    if (!Jx || !Ju || !res_c || (gterm && (!lam || !rho))) return DQP_ERR_BAD_ARG;
    AsmP P = {Jx, Ju, lam, res_c, rho, Jc, gterm, d->nbatch, d->n_state, d->n_ctrl, d->T, nullptr, nullptr, nullptr};
    DQP_LAUNCH(al_assemble_kernel, dim3(P.B), dim3(256), 0, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

__attribute__((visibility("default"))) int
dqp_al_merit(const dqp_al_mpc_dims *d, int32_t ncand, const double *xu, const double *x_next,
             const double *x0, const double *Qdiag, const double *q, const double *lam,
             const double *rho, const double *u_lower, const double *u_upper, double *merit, void *stream)
{
    if (!d || d->nbatch < 0 || d->n_state <= 0 || d->n_ctrl <= 0 || d->T < 2 || ncand < 0) return DQP_ERR_BAD_ARG;
    if (d->nbatch == 0 || ncand == 0) return DQP_OK;
    if (!xu || !x_next || !x0 || !Qdiag || !q || !lam || !rho || !u_lower || !u_upper || !merit)
        return DQP_ERR_BAD_ARG;
    MeritP P = {xu, x_next, x0, Qdiag, q, lam, rho, u_lower, u_upper, merit, d->nbatch, d->n_state,
                d->n_ctrl, d->T, ncand};
    const long long total = (long long)ncand * d->nbatch;
    DQP_LAUNCH(al_merit_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}


// workspace of dqp_al_newton_solve, in doubles per problem (see the carve below)
static size_t al_solve_doubles(int n, int m, int T)
{
    const size_t nt = n + m, nz = (size_t)T * nt, ncon = (size_t)T * n + 2 * (size_t)T * m;
    return (size_t)(T - 1) * n * n + (size_t)(T - 1) * n * m + ncon + ncon * nz + nz + nz + 20 + 1 + 1;
}

__attribute__((visibility("default"))) size_t dqp_al_newton_solve_bytes(const dqp_al_mpc_dims *d, int32_t banded)
{
    if (!d || d->nbatch <= 0 || d->n_state <= 0 || d->n_ctrl <= 0 || d->T < 2) return 0;
    if (banded)         // update + 20 candidate merits + current merit + info per problem
        return ((size_t)d->nbatch * ((size_t)d->T * (d->n_state + d->n_ctrl) + 20 + 1 + 1) + 2) * sizeof(double);
    return ((size_t)d->nbatch * al_solve_doubles(d->n_state, d->n_ctrl, d->T) + 2) * sizeof(double);
}

static int newton_solve_impl(const dqp_al_mpc_dims *d, int dyn_id, double dt, int32_t n_steps, int32_t banded,
                             const double *x0, const double *Qdiag, const double *q, const double *lam, const double *rho,
                             const double *u_lower, const double *u_upper, double *xu, double *L, double *status,
                             int32_t *fail, void *workspace, void *stream, bool clear_fail, bool keep_factor = true);

__attribute__((visibility("default"))) int
dqp_al_newton_solve(const dqp_al_mpc_dims *d, int dyn_id, double dt, int32_t n_steps, int32_t banded,
                    const double *x0, const double *Qdiag, const double *q, const double *lam, const double *rho,
                    const double *u_lower, const double *u_upper, double *xu, double *L, double *status,
                    int32_t *fail, void *workspace, void *stream)
{
    return newton_solve_impl(d, dyn_id, dt, n_steps, banded, x0, Qdiag, q, lam, rho, u_lower, u_upper, xu, L, status, fail,
                             workspace, stream, true);
}

// clear_fail = false: the caller's own kernel zeroed the flag (dqp_al_mpc_solve: al_start_kernel clears all of them)
static int newton_solve_impl(const dqp_al_mpc_dims *d, int dyn_id, double dt, int32_t n_steps, int32_t banded,
                             const double *x0, const double *Qdiag, const double *q, const double *lam, const double *rho,
                             const double *u_lower, const double *u_upper, double *xu, double *L, double *status,
                             int32_t *fail, void *workspace, void *stream, bool clear_fail, bool keep_factor)
{
    if (!d || d->nbatch < 0 || d->n_state <= 0 || d->n_ctrl <= 0 || d->T < 2 || n_steps < 1) return DQP_ERR_BAD_ARG;
    if (d->nbatch == 0) return DQP_OK;
    int32_t dn = 0, dm = 0;
    if (dqp_dyn_sizes(dyn_id, &dn, &dm) != DQP_OK || dn != d->n_state || dm != d->n_ctrl) return DQP_ERR_BAD_ARG;
    if (!x0 || !Qdiag || !q || !lam || !rho || !u_lower || !u_upper || !xu || !fail || !workspace) return DQP_ERR_BAD_ARG;
    const int B = d->nbatch, n = d->n_state, m = d->n_ctrl, T = d->T, nt = n + m, nz = T * nt;
    const int ncon = T * n + 2 * T * m;
    hipStream_t st = (hipStream_t)stream;
    double *w = (double *)workspace;
    if (banded) {
        // block-tridiagonal form (dqp_al_banded.hip): linearise + gradient + block Cholesky + solve in
        // ONE launch per step, no Jacobian / Hessian in HBM; L receives the banded factor
        if (nt > 16 || n > 12) return DQP_ERR_TOO_LARGE;
        if (!L) return DQP_ERR_BAD_ARG;
        double *upd = w;           w += (size_t)B * nz;
        double *merit = w;         w += (size_t)20 * B;
        double *merit_cur = w;     w += (size_t)B;
        int32_t *info = (int32_t *)w;
        if (clear_fail && hipMemsetAsync(fail, 0, sizeof(int32_t), st) != hipSuccess) return DQP_ERR_LAUNCH;
        LsAP Lp = {xu, upd, x0, Qdiag, q, lam, rho, u_lower, u_upper, merit_cur, dt, B, n, m, T, 0, dyn_id};
        int rc0 = launch_ls(Lp, st);                                    // merit at the start
        if (rc0) return rc0;
        for (int it = 0; it < n_steps; ++it) {
            // only the last step's factor is used afterwards (NewtonAL.backward, al_utils.py:477-480)
            int rc = dqp::al_banded_newton_step_keep(d, dyn_id, dt, xu, x0, Qdiag, q, lam, rho, u_lower, u_upper, upd, L,
                                                     info, stream, (it == n_steps - 1 && keep_factor) ? 1 : 0);
            if (rc) return rc;
            LsAP Lc = {xu, upd, x0, Qdiag, q, lam, rho, u_lower, u_upper, merit, dt, B, n, m, T, 20, dyn_id,
                       xu, merit_cur, status, fail, info};
            rc = launch_ls(Lc, st);
            if (rc != DQP_OK && rc != 2) return rc;
            if (rc != 2) {          // candidates split over several groups: select in a launch of its own
                SelP Se = {merit, upd, x0, xu, merit_cur, status, fail, info, B, n, nz, 20};
                DQP_LAUNCH(al_select_kernel, dim3(B), dim3(64), 0, st, Se);
            }
        }
        return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
    }
    if (d->n_state > 8 || d->n_ctrl > 2) return DQP_ERR_TOO_LARGE;
    dqp_al_dims ad = {B, nz, ncon, 0};
    AlP A = {};
    size_t lds = 0;
    int rc = fill(&ad, A, lds);
    if (rc) return rc;
    double *Jx = w;            w += (size_t)B * (T - 1) * n * n;
    double *Ju = w;            w += (size_t)B * (T - 1) * n * m;
    double *resc = w;          w += (size_t)B * ncon;
    double *Jc = w;            w += (size_t)B * ncon * nz;
    double *grad = w;          w += (size_t)B * nz;
    double *upd = w;           w += (size_t)B * nz;
    double *merit = w;         w += (size_t)20 * B;
    double *merit_cur = w;     w += (size_t)B;
    int32_t *info = (int32_t *)w;
    if (clear_fail && hipMemsetAsync(fail, 0, sizeof(int32_t), st) != hipSuccess) return DQP_ERR_LAUNCH;
    LsAP Lp = {xu, upd, x0, Qdiag, q, lam, rho, u_lower, u_upper, merit_cur, dt, B, n, m, T, 0, dyn_id};
    rc = launch_ls(Lp, st);                                             // merit at the start
    if (rc) return rc;
    for (int it = 0; it < n_steps; ++it) {
        LinP Li = {xu, x0, u_lower, u_upper, Jx, Ju, resc, dt, B, n, m, T, dyn_id};
        DQP_LAUNCH(al_linearize_kernel, dim3((unsigned)(((long long)B * T + 255) / 256)), dim3(256), 0, st, Li);
        AsmP As = {Jx, Ju, lam, resc, rho, Jc, grad, B, n, m, T, Qdiag, q, xu};
        DQP_LAUNCH(al_assemble_kernel, dim3(B), dim3(256), 0, st, As);
        AlP P = A;
        P.Jc = Jc; P.Qd = Qdiag; P.rho = rho; P.grad = grad; P.update = upd; P.L = L; P.info = info;
        switch ((nz + 15) / 16) {
        case 1: rc = launch(al_newton_kernel<1>, P, lds, stream); break;
        case 2: rc = launch(al_newton_kernel<2>, P, lds, stream); break;
        case 3: rc = launch(al_newton_kernel<3>, P, lds, stream); break;
        case 4: rc = launch(al_newton_kernel<4>, P, lds, stream); break;
        case 5: rc = launch(al_newton_kernel<5>, P, lds, stream); break;
        case 6: rc = launch(al_newton_kernel<6>, P, lds, stream); break;
        case 7: rc = launch(al_newton_kernel<7>, P, lds, stream); break;
        default: rc = launch(al_newton_kernel<8>, P, lds, stream); break;
        }
        if (rc) return rc;
        LsAP Lc = {xu, upd, x0, Qdiag, q, lam, rho, u_lower, u_upper, merit, dt, B, n, m, T, 20, dyn_id};
        rc = launch_ls(Lc, st);
        if (rc) return rc;
        SelP Se = {merit, upd, x0, xu, merit_cur, status, fail, info, B, n, nz, 20};
        DQP_LAUNCH(al_select_kernel, dim3(B), dim3(64), 0, st, Se);
    }
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}


__attribute__((visibility("default"))) int
dqp_al_outer_update(const dqp_al_mpc_dims *d, int dyn_id, double dt, const double *xu, const double *x0,
                    const double *lam, const double *rho, const double *Qdiag, const double *q,
                    const double *u_lower, const double *u_upper, double *lam_new, double *cost,
                    double *res_norm, void *stream)
{
    if (!d || d->nbatch < 0 || d->n_state <= 0 || d->n_ctrl <= 0 || d->T < 2) return DQP_ERR_BAD_ARG;
    if (d->nbatch == 0) return DQP_OK;
    int32_t dn = 0, dm = 0;
    if (dqp_dyn_sizes(dyn_id, &dn, &dm) != DQP_OK || dn != d->n_state || dm != d->n_ctrl) return DQP_ERR_BAD_ARG;
    if (d->n_state > 12 || d->n_state + d->n_ctrl > 16) return DQP_ERR_TOO_LARGE;
    if (!xu || !x0 || !lam || !rho || !Qdiag || !q || !u_lower || !u_upper || !lam_new || !cost || !res_norm)
        return DQP_ERR_BAD_ARG;
    OutP P = {xu, x0, lam, rho, Qdiag, q, u_lower, u_upper, lam_new, cost, res_norm, dt, d->nbatch, d->n_state,
              d->n_ctrl, d->T, dyn_id, nullptr};
    return launch_outer(P, (hipStream_t)stream);
}

__attribute__((visibility("default"))) size_t dqp_al_mpc_solve_bytes(const dqp_al_mpc_dims *d)
{
    return dqp_al_newton_solve_bytes(d, 1);
}

__attribute__((visibility("default"))) int
dqp_al_mpc_solve(const dqp_al_mpc_dims *d, int dyn_id, double dt, int32_t al_iter, int32_t newton_steps,
                 const double *x_init, const double *u_init, const double *x0, const double *Qdiag, const double *q,
                 const double *u_lower, const double *u_upper, const double *lam_in, const double *rho_in,
                 const double *prev_cost, const double *prev_lam, const double *prev_rho, int32_t n_prev,
                 double *xu, double *hist_cost, double *hist_lam, double *hist_rho, double *res_norm, double *factor,
                 double *status, int32_t *fail, void *workspace, void *stream)
{
    if (!d || d->nbatch < 0 || d->n_state <= 0 || d->n_ctrl <= 0 || d->T < 2 || al_iter < 1 || al_iter > 256 || newton_steps < 1 ||
        n_prev < 0)
        return DQP_ERR_BAD_ARG;
    if (d->nbatch == 0) return DQP_OK;
    if (!x_init || !u_init || !x0 || !Qdiag || !q || !u_lower || !u_upper || !lam_in || !rho_in || !xu || !hist_cost ||
        !hist_lam || !hist_rho || !res_norm || !factor || !fail || !workspace)
        return DQP_ERR_BAD_ARG;
    if (n_prev > 0 && (!prev_cost || !prev_lam || !prev_rho)) return DQP_ERR_BAD_ARG;
    if (d->n_state > 12 || d->n_state + d->n_ctrl > 16) return DQP_ERR_TOO_LARGE;
    const int B = d->nbatch, n = d->n_state, m = d->n_ctrl, T = d->T;
    const long long ncon = (long long)T * n + 2LL * T * m;
    hipStream_t st = (hipStream_t)stream;
    StartP S = {x_init, u_init, Qdiag, q, lam_in, rho_in, prev_cost, prev_lam, prev_rho, xu, hist_cost, hist_lam, hist_rho,
                B, n, m, T, n_prev, fail, al_iter};
    DQP_LAUNCH(al_start_kernel, dim3((unsigned)((B + 15) / 16)), dim3(256), 0, st, S);
    for (int i = 0; i < al_iter; ++i) {
        const double *lam = hist_lam + (long long)i * B * ncon, *rho = hist_rho + (long long)i * B;
        int rc = newton_solve_impl(d, dyn_id, dt, newton_steps, 1, x0, Qdiag, q, lam, rho, u_lower, u_upper, xu, factor,
                                   status, fail + i, workspace, stream, false, i == al_iter - 1);
        if (rc) return rc;
        OutP O = {xu, x0, lam, rho, Qdiag, q, u_lower, u_upper, hist_lam + (long long)(i + 1) * B * ncon,
                  hist_cost + (long long)(i + 1) * B, res_norm, dt, B, n, m, T, dyn_id, hist_rho + (long long)(i + 1) * B};
        if ((rc = launch_outer(O, st)) != DQP_OK) return rc;
    }
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // extern "C"
