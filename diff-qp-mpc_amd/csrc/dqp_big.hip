// dqp_big.hip -- dense QPs above DQP_MAX_DIM (64 < max(nz, nineq, neq) <= DQP_MAX_DIM_LARGE = 512): the reference's own
// profiler sizes (prof-linear.py:38-46: nz = nineq in {10, 50, 100, 500}) and the l1-slack reformulation of an MPC QP
// (sl1qp_mpc.py:703-752: nz + 2 neq + nineq variables).  One QP per 256-thread workgroup; same Mehrotra iteration,
// same coordinates and the same block elimination as the one-wavefront kernels of dqp_pdipm.hip
// (batch.py:46-208 / 351-469):
//
//   xh = Lq^T x (Q = Lq Lq^T),  Gh = G Lq^-T,  Ah = A Lq^-T,  S1 = Ah Ah^T = L1 L1^T,  GA = Gh Ah^T L1^-T,
//   R = Gh Gh^T - GA GA^T   (the reference's Schur complement, batch.py:399,420),   T = R + diag(s/z) per iteration.
//
// The matrices live in the caller's workspace (dqp_workspace_bytes; ~10 MB per QP at nz = nineq = 500), padded to
// multiples of 64 with an identity on the padded diagonal, and every O(n^3) operation is a sequence of 64 x 64 output
// tiles  C -= X Y^T  accumulated on the fp64 matrix cores (v_mfma_f64_16x16x4_f64; each wavefront owns a 32 x 32
// quadrant = 2 x 2 MFMA tiles, operands staged through LDS in 32-column chunks):
//   * potrf: left-looking blocked Cholesky -- tile (i, j) = A_ij - sum_{k<j} L_ik L_jk^T, the diagonal tile is
//     factored in LDS and its inverse kept (dinv), an off-diagonal tile is multiplied by dinv_j^T (again an X Y^T tile);
//   * trsm (B <- B L^-T) and the Gram products Gh Gh^T, Ah Ah^T, Gh Ah^T: the same tile routine;
//   * triangular solves with vectors use the stored inverses of the diagonal blocks (all mat-vecs, no 64-step chains).
// Vectors live in LDS.  Termination: per problem, or the batch rule's pass 1 (history) / pass 2 (re-solve of the
// flagged problems up to the batch's stop) like the generic kernels.
//
// Reference functions covered: a2-a8 of SURVEY.md section 8 at the sizes the one-wavefront kernels do not reach.
#include <hip/hip_runtime.h>
#include <math.h>

#include "dqp_common.h"

namespace dqp {
namespace big {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int TB = 64;          // tile edge
constexpr int KC = 32;          // K chunk staged in LDS
constexpr int LK = KC + 1;      // row stride of a staged chunk
constexpr int LT = TB + 1;      // row stride of the LDS tile
constexpr int NTHR = 256;
constexpr int DB = 2 * TB * TB; // doubles per diagonal block in a `dinv` array: the inverse, then its transpose

__host__ __device__ inline int pad64(int n) { return (n + TB - 1) / TB * TB; }

// workspace layout per QP (doubles); every matrix row-major with its padded column count as leading dimension
struct Lay {
    int NP, MP, EP;
    long long oLq, oLqi, oGh, oAh, oL1, oL1i, oGA, oR, oT, oTi, oBest, oProf, total;
};
__host__ __device__ inline Lay layout(int N, int M, int E)
{
    Lay L;
    L.NP = pad64(N); L.MP = pad64(M); L.EP = E > 0 ? pad64(E) : 0;
    long long o = 0;
    L.oLq = o;  o += (long long)L.NP * L.NP;
    L.oLqi = o; o += 2LL * L.NP * TB;               // inverses of the diagonal blocks of Lq, and their transposes
    L.oGh = o;  o += (long long)L.MP * L.NP;
    L.oAh = o;  o += (long long)L.EP * L.NP;
    L.oL1 = o;  o += (long long)L.EP * L.EP;
    L.oL1i = o; o += 2LL * L.EP * TB;
    L.oGA = o;  o += (long long)L.MP * L.EP;
    L.oR = o;   o += (long long)L.MP * L.MP;
    L.oT = o;   o += (long long)L.MP * L.MP;
    L.oTi = o;  o += 2LL * L.MP * TB;
    L.oBest = o; o += (long long)L.NP + 2 * L.MP + L.EP;
    L.oProf = o; o += 16;                           // -DDQP_BIG_PROF: cycles per phase (tools/bench_big.py)
    L.total = o;
    return L;
}

// LDS carve-up as OFFSETS (doubles) into the kernel's dynamic LDS: the helpers below are real functions (__noinline__:
// inlined, the kernel is one 40 k-instruction body), and an LDS pointer handed through a call is a generic pointer --
// every access a flat_load / flat_store the compiler can neither pipeline nor tell apart from global memory.  With
// offsets and the base as an address-space-3 pointer every function keeps ds_ instructions.
typedef int lptr;
struct Sh {
    lptr xs, ys, ct, red;
    lptr xh, s, z, y, ph, hh, bb, rxh, rz, ry, tn, tn2, g, dsa, dz, ds, tm, te, te2, dxh;
};
// The LDS base travels as an explicit address-space-3 pointer argument.  (Declaring the dynamic LDS array inside the
// helpers works too, but a non-kernel function then finds it through llvm.amdgcn.dynlds.offset.table indexed by a kernel
// id in s15; with this compiler some builds of this file read a clobbered id there -- garbage LDS offsets, wrong
// results in one build and correct ones in the next, same source.)
typedef __attribute__((address_space(3))) double *LP;
// (the base goes through an empty asm: interprocedural constant propagation would otherwise put the array's address back
// into the helpers, and with it the table lookup)
__device__ __forceinline__ LP lds_base()
{
    extern __shared__ __attribute__((aligned(16))) double dqp_big_lds[];
    unsigned base = (unsigned)(unsigned long long)(LP)dqp_big_lds;
    asm volatile("" : "+s"(base));
    return (LP)(unsigned long long)base;
}
__host__ __device__ inline size_t lds_doubles(const Lay &L)
{
    return 2 * (size_t)TB * LK + (size_t)TB * LT + 5 * TB + 16 + 2 * TB + 5 * (size_t)L.NP + 9 * (size_t)L.MP +
           5 * (size_t)(L.EP > 0 ? L.EP : TB);
}
__host__ __device__ inline Sh carve(const Lay &L)
{
    Sh S;
    int p = 0;
    const int EPv = L.EP > 0 ? L.EP : TB;
    S.xs = p; p += TB * LK; S.ys = p; p += TB * LK;
    S.ct = p; p += TB * LT + 5 * TB;            // the tile; doubles as the vector kernels' scratch
    S.red = p; p += 16 + 2 * TB;
    S.xh = p; p += L.NP; S.ph = p; p += L.NP; S.rxh = p; p += L.NP; S.tn = p; p += L.NP; S.tn2 = p; p += L.NP;
    S.dxh = S.tn2;          // kkt_xy builds the x-direction in place of its own temporary
    S.s = p; p += L.MP; S.z = p; p += L.MP; S.hh = p; p += L.MP; S.rz = p; p += L.MP; S.g = p; p += L.MP;
    S.dsa = p; p += L.MP; S.dz = p; p += L.MP; S.ds = p; p += L.MP; S.tm = p; p += L.MP;
    S.y = p; p += EPv; S.bb = p; p += EPv; S.ry = p; p += EPv; S.te = p; p += EPv; S.te2 = p; p += EPv;
    return S;
}

// Everything a helper needs is passed BY VALUE (three sizes and the workspace pointer; the layouts are recomputed):
// a struct handed to a __noinline__ function by reference lives in the caller's scratch and is read back through flat
// scratch addressing.
struct Dims { int N, M, E; };
__device__ __forceinline__ Sh carve(Dims D) { return carve(layout(D.N, D.M, D.E)); }

// ---------------------------------------------------------------------------------------------------------------
// workgroup reductions (256 threads, four wavefronts)
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wg_sum(LP sm, double v, lptr redo)
{
    LP red = sm + redo;
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ double wg_min(LP sm, double v, lptr redo)
{
    LP red = sm + redo;
    v = wave_min(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
}

// ---------------------------------------------------------------------------------------------------------------
// acc += X[0..64)[k0..k1) * Y[0..64)[k0..k1)^T : X, Y row-major in global memory (or X an LDS tile with stride LT);
// wavefront w owns rows 32 (w >> 1) .. and columns 32 (w & 1) .. of the 64 x 64 tile as 2 x 2 MFMA tiles.
// f64 16x16x4 MFMA: A operand lane l = A[l & 15][l >> 4], B operand lane l = B[l >> 4][l & 15],
// C/D: column l & 15, row (l >> 4) + 4 reg.
// XLDS: the X operand is the LDS tile at offset xl (stride LT) instead of global memory
template <bool XLDS>
__device__ __forceinline__ void tile_mm(LP sm, const double *X, lptr xl, int ldx, const double *Y, int ldy, int k0, int k1,
                                        double4_t (&acc)[2][2], const Sh &S)
{
    constexpr int PT = TB * KC / NTHR;      // staged elements per thread and operand
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int ra = 32 * (w >> 1) + l15, rb = 32 * (w & 1) + l15;
    if (k0 >= k1) return;
    // the next chunk's global loads are issued before the MFMAs of the current one (registers as the second buffer)
    double px[PT], py[PT];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int idx = tid + NTHR * u, r = idx >> 5, c = idx & 31;
            if (!XLDS) px[u] = X[(long long)r * ldx + kc + c];
            py[u] = Y[(long long)r * ldy + kc + c];
        }
    };
    fetch(k0);
    for (int kc = k0; kc < k1; kc += KC) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int idx = tid + NTHR * u, r = idx >> 5, c = idx & 31;
            if (!XLDS) sm[S.xs + r * LK + c] = px[u];
            sm[S.ys + r * LK + c] = py[u];
        }
        __syncthreads();
        if (kc + KC < k1) fetch(kc + KC);
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            double a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = XLDS ? sm[xl + (ra + 16 * t) * LT + (kc - k0) + 4 * ks + kq] : sm[S.xs + (ra + 16 * t) * LK + 4 * ks + kq];
                b[t] = sm[S.ys + (rb + 16 * t) * LK + 4 * ks + kq];
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void acc_zero(double4_t (&acc)[2][2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
}

// visit the 16 accumulator elements of this lane: f(row in tile, column in tile, value)
template <class F> __device__ __forceinline__ void acc_each(const double4_t (&acc)[2][2], F f)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                f(32 * (w >> 1) + 16 * ti + kq + 4 * rg, 32 * (w & 1) + 16 * tj + l15, acc[ti][tj][rg]);
}

// Cholesky of the 64 x 64 LDS tile ct (lower, in place, zero upper part); false if a pivot is not positive.
// Thread (w, lane) keeps the 16 elements (w + 4u, lane), u < 16, in registers; per step the owners of column k publish it
// (unscaled) in one of two LDS column buffers, one barrier, and every thread applies the rank-1 update to its registers.
// (The first version updated the tile in LDS: every element was a load-load-load-fma-store chain the compiler could not
// pipeline -- possible aliasing -- 320 k cycles per tile against 25 k.)
__device__ __noinline__ bool chol64(LP sm, lptr cto, lptr colo, double minpiv)
{
    LP ct = sm + cto, colbuf = sm + colo;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = ct[(w + 4 * u) * LT + lane];
    bool ok = true;
#pragma nounroll    // (the build's unroll thresholds are set for the register-resident kernels: unrolled, this loop is 30 k
                    // instructions of straight-line code and runs at instruction-fetch speed)
    for (int k = 0; k < TB; ++k) {
        LP cb = colbuf + (k & 1) * TB;
        if (lane == k) {
#pragma unroll
            for (int u = 0; u < 16; ++u) cb[w + 4 * u] = a[u];
        }
        __syncthreads();
        const double d = cb[k], lc = cb[lane];                      // pivot, (unscaled) L[lane][k]
        double li[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) li[u] = cb[w + 4 * u];         // all LDS reads of the step in flight at once
        if (!(d > minpiv)) ok = false;
        const double dd = d > minpiv ? d : 1.0;
        double r = __builtin_amdgcn_rsq(dd);                        // hardware estimate + two Newton steps
        r = r * fma(-0.5 * dd * r, r, 1.5);
        r = r * fma(-0.5 * dd * r, r, 1.5);
        const double r2lc = r * r * lc;
        if (lane == k) {                                            // column k becomes final
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] = (w + 4 * u) >= k ? li[u] * r : 0.0;
        } else if (lane > k) {                                      // trailing update of the lower triangle
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const double upd = fma(-li[u], r2lc, a[u]);
                a[u] = (w + 4 * u) >= lane ? upd : a[u];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int i = w + 4 * u;
        ct[i * LT + lane] = lane <= i ? a[u] : 0.0;
    }
    __syncthreads();
    return ok;
}

// inv = L^-1 for the lower-triangular LDS tile L (stride LT) -> `inv` (LDS, stride LT).  Thread c of the first
// wavefront builds column c by forward substitution with the column in registers (fully unrolled: every L[i][k] is one
// broadcast LDS read for the whole wavefront; a version with the column in LDS was a chain of 2016 dependent LDS round
// trips per thread, 110 us per block).
__device__ __noinline__ void inv64(LP sm, lptr Lo, lptr invo)
{
    LP L = sm + Lo, inv = sm + invo;
    const int c = threadIdx.x;
    if (c < TB) {
        double x[TB];
        const double rdc = 1.0 / L[c * LT + c];
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            // four partial sums: the chain of i dependent FMAs of one row becomes i / 4
            double s0 = (i == c) ? 1.0 : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int k = 0; k + 3 < i; k += 4) {
                s0 = fma(-L[i * LT + k], x[k], s0);
                s1 = fma(-L[i * LT + k + 1], x[k + 1], s1);
                s2 = fma(-L[i * LT + k + 2], x[k + 2], s2);
                s3 = fma(-L[i * LT + k + 3], x[k + 3], s3);
            }
#pragma unroll
            for (int k = i & ~3; k < i; ++k) s0 = fma(-L[i * LT + k], x[k], s0);
            const double rdi = __shfl(rdc, i, 64);              // 1 / L[i][i]: thread i's own reciprocal
            x[i] = i < c ? 0.0 : ((s0 + s1) + (s2 + s3)) * rdi;
        }
#pragma unroll
        for (int i = 0; i < TB; ++i) inv[i * LT + c] = x[i];
    }
    __syncthreads();
}

#ifdef DQP_BIG_PROF
#define SUBT0() unsigned long long _st = __builtin_readcyclecounter()
#define SUB(k) do { const unsigned long long _n = __builtin_readcyclecounter(); if (threadIdx.x == 0) sm[S.red + 8 + (k)] += (double)(_n - _st); _st = _n; } while (0)
#else
#define SUBT0()
#define SUB(k)
#endif

// A (n x n, n a multiple of 64, row-major, ld) <- its lower Cholesky factor (upper part zeroed); the inverses of
// the diagonal blocks go to dinv (n/64 blocks of 64 x 64, row-major, ld 64).  Returns false on a non-positive pivot.
__device__ __noinline__ bool potrf(LP sm, double *A, int ld, int n, double *dinv, Dims D, double minpiv)
{
    const Sh S = carve(D);
    const int nb = n / TB, tid = threadIdx.x;
    bool ok = true;
    const lptr invo = S.xs;         // 64 x 65 <= 2 x 64 x 33: the staging area doubles as the inverse buffer
    const LP inv = sm + invo;
    for (int jb = 0; jb < nb; ++jb) {
        for (int ib = jb; ib < nb; ++ib) {
            double4_t acc[2][2];
            acc_zero(acc);
            double *Aij = A + (long long)ib * TB * ld + jb * TB;
            SUBT0();
            tile_mm<false>(sm, A + (long long)ib * TB * ld, 0, ld, A + (long long)jb * TB * ld, ld, 0, jb * TB, acc, S);
            __syncthreads();
            SUB(0);
            acc_each(acc, [&](int r, int c, double v) { sm[S.ct + r * LT + c] = Aij[(long long)r * ld + c] - v; });
            __syncthreads();
            SUB(1);
            if (ib == jb) {
                if (!chol64(sm, S.ct, S.red + 16, minpiv)) ok = false;
                SUB(2);
                inv64(sm, S.ct, invo);
                SUB(3);
#pragma unroll 4
                for (int idx = tid; idx < TB * TB; idx += NTHR) {
                    const int r = idx >> 6, c = idx & 63;
                    Aij[(long long)r * ld + c] = sm[S.ct + r * LT + c];
                    dinv[(long long)jb * DB + idx] = inv[r * LT + c];
                    dinv[(long long)jb * DB + TB * TB + idx] = inv[c * LT + r];        // the transpose, for trsv_L
                }
                // the rest of the block row (the upper triangle) is zero
                {
                    const int lane = tid & 63, w = tid >> 6;
                    for (int r = w; r < TB; r += 4)
                        for (int c = (jb + 1) * TB + lane; c < n; c += 64) A[(long long)(jb * TB + r) * ld + c] = 0.0;
                }
                __syncthreads();
                SUB(4);
            } else {
                double4_t a2[2][2];
                acc_zero(a2);
                tile_mm<true>(sm, nullptr, S.ct, LT, dinv + (long long)jb * DB, TB, 0, TB, a2, S);      // C dinv_j^T
                acc_each(a2, [&](int r, int c, double v) { Aij[(long long)r * ld + c] = v; });
                __syncthreads();
                SUB(5);
            }
        }
    }
    return ok;
}

// B (m x n, both multiples of 64) <- B L^-T for the factor L (n x n) and the inverses of its diagonal blocks
__device__ __noinline__ void trsm_rlt(LP sm, double *B, int ldb, int m, const double *Lf, int ldl, int n, const double *dinv, Dims D)
{
    const Sh S = carve(D);
    for (int jb = 0; jb < n / TB; ++jb)
        for (int ib = 0; ib < m / TB; ++ib) {
            double4_t acc[2][2], a2[2][2];
            acc_zero(acc); acc_zero(a2);
            double *Bij = B + (long long)ib * TB * ldb + jb * TB;
            tile_mm<false>(sm, B + (long long)ib * TB * ldb, 0, ldb, Lf + (long long)jb * TB * ldl, ldl, 0, jb * TB, acc, S);
            __syncthreads();
            acc_each(acc, [&](int r, int c, double v) { sm[S.ct + r * LT + c] = Bij[(long long)r * ldb + c] - v; });
            __syncthreads();
            tile_mm<true>(sm, nullptr, S.ct, LT, dinv + (long long)jb * DB, TB, 0, TB, a2, S);
            acc_each(a2, [&](int r, int c, double v) { Bij[(long long)r * ldb + c] = v; });
            __syncthreads();
        }
}

// C (m x n) = beta C + alpha X Y^T, X (m x K), Y (n x K), all multiples of 64; lower: only tiles with i >= j
__device__ __noinline__ void gemm_nt(LP sm, double *C, int ldc, const double *X, int ldx, const double *Y, int ldy, int m, int n, int K,
                        double alpha, double beta, bool lower, Dims D)
{
    const Sh S = carve(D);
    for (int ib = 0; ib < m / TB; ++ib)
        for (int jb = 0; jb < (lower ? ib + 1 : n / TB); ++jb) {
            double4_t acc[2][2];
            acc_zero(acc);
            tile_mm<false>(sm, X + (long long)ib * TB * ldx, 0, ldx, Y + (long long)jb * TB * ldy, ldy, 0, K, acc, S);
            double *Cij = C + (long long)ib * TB * ldc + jb * TB;
            acc_each(acc, [&](int r, int c, double v) {
                const double o = beta != 0.0 ? beta * Cij[(long long)r * ldc + c] : 0.0;
                Cij[(long long)r * ldc + c] = fma(alpha, v, o);
            });
        }
    __syncthreads();
}

// mirror the lower triangle of C (n x n) into the upper one
__device__ __noinline__ void symmetrize(double *C, int ld, int n)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int j = w; j < n; j += 4)              // row j of the lower triangle -> column j of the upper one
        for (int i = lane; i < j; i += 64) C[(long long)i * ld + j] = C[(long long)j * ld + i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// vector kernels (vectors in LDS; x and y must not alias)
// Dot products of up to 16 consecutive rows with x over columns [0, cols), by ONE wavefront: lanes along the (contiguous)
// columns, sixteen accumulators per lane (sixteen independent loads in flight), then a transpose through this
// wavefront's LDS scratch (16 x 65 doubles).  Returns, on every lane, the total of row (lane & 15).
__device__ __forceinline__ double wave_dot16(LP sm, const double *rowp, int ld, int nr, int cols, lptr xo, lptr scro)
{
    const LP x = sm + xo, scr = sm + scro;
    const int lane = threadIdx.x & 63;
    double a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = 0.0;
    for (int c = lane; c < cols; c += 64) {
        const double xv = x[c];
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = u < nr ? rowp[(long long)u * ld + c] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = fma(v[u], xv, a[u]);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) scr[u * 65 + lane] = a[u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int u = lane & 15, q = lane >> 4;
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) p += scr[u * 65 + 16 * q + k];
    p += __shfl_xor(p, 16, 64);
    p += __shfl_xor(p, 32, 64);
    __builtin_amdgcn_wave_barrier();
    return p;
}
// y[0..rows) = M x, M (rows x cols) row-major; scr: 4 x 16 x 65 doubles of LDS
__device__ __noinline__ void matvec(LP sm, lptr yo, const double *Mx, int ld, int rows, int cols, lptr x, lptr scr)
{
    const LP y = sm + yo;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r0 = 16 * w; r0 < rows; r0 += 64) {
        const int nr = min(16, rows - r0);
        const double v = wave_dot16(sm, Mx + (long long)r0 * ld, ld, nr, cols, x, scr + w * 16 * 65);
        if (lane < nr) y[r0 + lane] = v;
    }
    __syncthreads();
}
// y[0..cols) = M^T x: lanes along the (contiguous) columns, the four wavefronts split the rows (eight loads in flight per
// lane), partial sums combined through LDS scratch (4 x 64 doubles at `scr`)
__device__ __noinline__ void matvecT(LP sm, lptr yo, const double *Mx, int ld, int rows, int cols, lptr xo, lptr scro)
{
    const LP y = sm + yo, scr = sm + scro, x = sm + xo;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c0 = 0; c0 < cols; c0 += 64) {
        const int c = c0 + lane;
        const bool in = c < cols;
        double a = 0.0;
        int r = w;
        for (; r + 28 < rows; r += 32) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in ? Mx[(long long)(r + 4 * u) * ld + c] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fma(v[u], x[r + 4 * u], a);
        }
        for (; r < rows; r += 4) a = fma(in ? Mx[(long long)r * ld + c] : 0.0, x[r], a);
        __syncthreads();
        scr[w * 64 + lane] = a;
        __syncthreads();
        if (w == 0 && in) y[c] = scr[lane] + scr[64 + lane] + scr[128 + lane] + scr[192 + lane];
    }
    __syncthreads();
}
// x <- L^-1 x  (forward), with the inverses of the diagonal blocks (dinv: [block][inverse | transpose]); tmp: the LDS tile
__device__ __noinline__ void trsv_L(LP sm, lptr xo, const double *Lf, int ld, int n, const double *dinv, lptr tmpo)
{
    const LP x = sm + xo, tmp = sm + tmpo;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const LP t = tmp + 4 * 16 * 65, part = t + TB;
    for (int jb = 0; jb < n / TB; ++jb) {
        // t = b_j - L[j, :j] x[:j]: wavefront w takes rows 16 w .. 16 w + 15 of the block
        const double v = wave_dot16(sm, Lf + (long long)(jb * TB + 16 * w) * ld, ld, 16, jb * TB, xo, tmpo + w * 16 * 65);
        if (lane < 16) t[16 * w + lane] = x[jb * TB + 16 * w + lane] - v;
        __syncthreads();
        // x_j = dinv_j t: thread (r, q) sums columns 16 q .. 16 q + 15 of row r from the transposed inverse (coalesced)
        const double *dT = dinv + (long long)jb * DB + TB * TB;
        double p = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) p = fma(dT[(16 * w + k) * TB + lane], t[16 * w + k], p);
        part[w * TB + lane] = p;
        __syncthreads();
        if (tid < TB) x[jb * TB + tid] = part[tid] + part[TB + tid] + part[2 * TB + tid] + part[3 * TB + tid];
        __syncthreads();
    }
}
// x <- L^-T x  (backward)
__device__ __noinline__ void trsv_LT(LP sm, lptr xo, const double *Lf, int ld, int n, const double *dinv, lptr tmpo)
{
    const LP x = sm + xo, tmp = sm + tmpo;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const LP t = tmp, part = tmp + TB;
    for (int jb = n / TB - 1; jb >= 0; --jb) {
        // t = b_j - L[j+1:, j]^T x[j+1:]: lanes along the 64 columns of the block, the wavefronts split the rows
        double a = 0.0;
        int r = (jb + 1) * TB + w;
        for (; r + 28 < n; r += 32) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Lf[(long long)(r + 4 * u) * ld + jb * TB + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fma(v[u], x[r + 4 * u], a);
        }
        for (; r < n; r += 4) a = fma(Lf[(long long)r * ld + jb * TB + lane], x[r], a);
        __syncthreads();
        part[w * TB + lane] = a;
        __syncthreads();
        if (tid < TB) t[tid] = x[jb * TB + tid] - (part[tid] + part[TB + tid] + part[2 * TB + tid] + part[3 * TB + tid]);
        __syncthreads();
        // x_j = dinv_j^T t: x_j[r] = sum_c dinv[c][r] t[c], rows of dinv contiguous along r
        const double *dI = dinv + (long long)jb * DB;
        double p = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) p = fma(dI[(16 * w + k) * TB + lane], t[16 * w + k], p);
        __syncthreads();
        part[w * TB + lane] = p;
        __syncthreads();
        if (tid < TB) x[jb * TB + tid] = part[tid] + part[TB + tid] + part[2 * TB + tid] + part[3 * TB + tid];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct Ctx {
    Dims D;
    Lay L;
    Sh S;
    double *ws;
    int N, M, E;
    double *Lq, *Lqi, *Gh, *Ah, *L1, *L1i, *GA, *R, *T, *Ti;
};
__device__ __forceinline__ Ctx ctx_of(double *ws, Dims D)
{
    const Lay L = layout(D.N, D.M, D.E);
    return Ctx{D, L, carve(L), ws, D.N, D.M, D.E, ws + L.oLq, ws + L.oLqi, ws + L.oGh, ws + L.oAh, ws + L.oL1,
               ws + L.oL1i, ws + L.oGA, ws + L.oR, ws + L.oT, ws + L.oTi};
}

// Lq, Gh, Ah, L1, GA, R from (Q, G, A).  Returns the status.
__device__ __noinline__ int setup(LP sm, double *ws, Dims D, const double *Q, const double *G, const double *A)
{
    const Ctx C = ctx_of(ws, D);
    const Lay &L = C.L;
    const int N = C.N, M = C.M, E = C.E, NP = L.NP, MP = L.MP, EP = L.EP, tid = threadIdx.x;
    int status = DQP_STATUS_OK;
    // padded copies: identity on the padded diagonal of Q, zero rows / columns elsewhere (a wavefront per row)
    const int lane = tid & 63, w = tid >> 6;
    auto copy_padded = [&](double *dst, int rowsP, const double *src, int rows, bool eye) {
        for (int i0 = w; i0 < rowsP; i0 += 32)
            for (int j = lane; j < NP; j += 64) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 4 * u;
                    v[u] = (i < rows && j < N) ? src[(long long)i * N + j] : ((eye && i == j) ? 1.0 : 0.0);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 4 * u < rowsP) dst[(long long)(i0 + 4 * u) * NP + j] = v[u];
            }
    };
    copy_padded(C.Lq, NP, Q, N, true);
    copy_padded(C.Gh, MP, G, M, false);
    if (E > 0) copy_padded(C.Ah, EP, A, E, false);
    __syncthreads();
    if (!potrf(sm, C.Lq, NP, NP, C.Lqi, D, 0.0)) status = DQP_STATUS_Q_NOT_PD;
    trsm_rlt(sm, C.Gh, NP, MP, C.Lq, NP, NP, C.Lqi, D);
    gemm_nt(sm, C.R, MP, C.Gh, NP, C.Gh, NP, MP, MP, NP, 1.0, 0.0, true, D);                      // R = Gh Gh^T
    if (E > 0) {
        trsm_rlt(sm, C.Ah, NP, EP, C.Lq, NP, NP, C.Lqi, D);
        gemm_nt(sm, C.L1, EP, C.Ah, NP, C.Ah, NP, EP, EP, NP, 1.0, 0.0, true, D);                 // S1 = Ah Ah^T
        for (int i = E + tid; i < EP; i += NTHR) C.L1[(long long)i * EP + i] = 1.0;             // padded diagonal
        __syncthreads();
        if (!potrf(sm, C.L1, EP, EP, C.L1i, D, 1e-13) && status == DQP_STATUS_OK) status = DQP_STATUS_A_RANK_DEF;
        gemm_nt(sm, C.GA, EP, C.Gh, NP, C.Ah, NP, MP, EP, NP, 1.0, 0.0, false, D);                // Gh Ah^T
        trsm_rlt(sm, C.GA, EP, MP, C.L1, EP, EP, C.L1i, D);                                       // ... L1^-T
        gemm_nt(sm, C.R, MP, C.GA, EP, C.GA, EP, MP, MP, EP, -1.0, 1.0, true, D);                 // R -= GA GA^T
    }
    symmetrize(C.R, MP, MP);
    return status;
}

// T = R + diag(dinv) (identity on the padded diagonal), factored; false on a non-positive pivot
__device__ __noinline__ bool factor_T(LP sm, double *ws, Dims D, lptr dinvo)
{
    const Ctx C = ctx_of(ws, D);
    const LP dinv = sm + dinvo;
    const int MP = C.L.MP, M = C.M, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // eight rows per thread and pass: the loads of a pass are all in flight before its first store
    for (int i0 = w; i0 < MP; i0 += 32)
        for (int j = lane; j < MP; j += 64) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 4 * u;
                v[u] = (i < MP && j <= i) ? C.R[(long long)i * MP + j] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 4 * u;
                if (i < MP) C.T[(long long)i * MP + j] = i == j ? (i < M ? v[u] + dinv[i] : 1.0) : v[u];
            }
        }
    __syncthreads();
    return potrf(sm, C.T, MP, MP, C.Ti, D, 0.0);
}

// S1^-1 v in place (E-space), through L1
__device__ __noinline__ void solve_S1(LP sm, double *ws, Dims D, lptr v)
{
    const Ctx C = ctx_of(ws, D);
    trsv_L(sm, v, C.L1, C.L.EP, C.L.EP, C.L1i, C.S.ct);
    trsv_LT(sm, v, C.L1, C.L.EP, C.L.EP, C.L1i, C.S.ct);
}

// z-part of solve_kkt in hat coordinates (dqp_pdipm.hip: kkt_wz): dz -> out (M-space).  rsd = rs / d.
//   u = rxh - Ah^T S1^-1 (Ah rxh - ry);   g = rz - rsd - Gh u;   dz = T^-1 g
__device__ __noinline__ void kkt_wz(LP sm, double *ws, Dims D, lptr rxho, lptr rsdo, lptr rzo, lptr ryo, lptr outo)
{
    const Ctx C = ctx_of(ws, D);
    const Sh &S = C.S;
    const LP rxh = sm + rxho, rsd = sm + (rsdo >= 0 ? rsdo : 0), rz = sm + rzo, ry = sm + ryo, out = sm + outo;
    const int N = C.N, M = C.M, E = C.E, NP = C.L.NP, MP = C.L.MP, EP = C.L.EP;
    lptr u = rxho;
    if (E > 0) {
        matvec(sm, S.te, C.Ah, NP, E, N, rxho, S.ct);
        for (int i = threadIdx.x; i < EP; i += NTHR) sm[S.te + i] = i < E ? sm[S.te + i] - ry[i] : 0.0;
        __syncthreads();
        solve_S1(sm, ws, D, S.te);
        matvecT(sm, S.tn, C.Ah, NP, E, N, S.te, S.ct);
        for (int i = threadIdx.x; i < N; i += NTHR) sm[S.tn + i] = rxh[i] - sm[S.tn + i];
        __syncthreads();
        u = S.tn;
    }
    matvec(sm, S.tm, C.Gh, NP, M, N, u, S.ct);
    for (int i = threadIdx.x; i < MP; i += NTHR) out[i] = i < M ? rz[i] - (rsdo >= 0 ? rsd[i] : 0.0) - sm[S.tm + i] : 0.0;
    __syncthreads();
    trsv_L(sm, outo, C.T, MP, MP, C.Ti, S.ct);
    trsv_LT(sm, outo, C.T, MP, MP, C.Ti, S.ct);
}

// x / y part for the direction dz (kkt_xy): q = rxh + Gh^T dz; e = S1^-1 (Ah q - ry); dxh = -q + Ah^T e; dy = -e
__device__ __noinline__ void kkt_xy(LP sm, double *ws, Dims D, lptr rxho, lptr ryo, lptr dzo, lptr dxho, lptr dyo)
{
    const Ctx C = ctx_of(ws, D);
    const Sh &S = C.S;
    const LP rxh = sm + rxho, ry = sm + (ryo >= 0 ? ryo : 0), dxh = sm + dxho, dy = sm + dyo;
    const int N = C.N, M = C.M, E = C.E, NP = C.L.NP, EP = C.L.EP;
    matvecT(sm, S.tn2, C.Gh, NP, M, N, dzo, S.ct);
    for (int i = threadIdx.x; i < N; i += NTHR) sm[S.tn2 + i] += rxh[i];
    __syncthreads();
    if (E > 0) {
        matvec(sm, S.te, C.Ah, NP, E, N, S.tn2, S.ct);
        for (int i = threadIdx.x; i < EP; i += NTHR) sm[S.te + i] = i < E ? sm[S.te + i] - (ryo >= 0 ? ry[i] : 0.0) : 0.0;
        __syncthreads();
        solve_S1(sm, ws, D, S.te);
        matvecT(sm, S.tn, C.Ah, NP, E, N, S.te, S.ct);
        for (int i = threadIdx.x; i < N; i += NTHR) dxh[i] = sm[S.tn + i] - sm[S.tn2 + i];
        for (int i = threadIdx.x; i < E; i += NTHR) dy[i] = -sm[S.te + i];
    } else {
        for (int i = threadIdx.x; i < N; i += NTHR) dxh[i] = -sm[S.tn2 + i];
    }
    __syncthreads();
}

__device__ __forceinline__ Ctx make_ctx(const KParams &P, long long qp)
{
    const Dims D = {P.N, P.M, P.E};
    return ctx_of(P.workspace + qp * layout(P.N, P.M, P.E).total, D);
}

#ifdef DQP_BIG_PROF
#define PROF_T0() unsigned long long _t = __builtin_readcyclecounter(); double _acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(k) do { const unsigned long long _n = __builtin_readcyclecounter(); _acc[k] += (double)(_n - _t); _t = _n; } while (0)
#define PROF_OUT() do { if (threadIdx.x == 0) for (int _k = 0; _k < 8; ++_k) { (C.ws + C.L.oProf)[_k] = _acc[_k]; (C.ws + C.L.oProf)[8 + _k] = sm[C.S.red + 8 + _k]; } } while (0)
#else
#define PROF_T0()
#define PROF(k)
#define PROF_OUT()
#endif

__global__ __launch_bounds__(NTHR) void forward_kernel(KParams P)
{
    const LP sm = lds_base();
    const long long qp = blockIdx.x;
    const int tid = threadIdx.x;
    term_zero_acc(P);
    int maxIter = P.maxIter;
    const bool batch = (P.flags & DQP_FLAG_BATCH_TERMINATION) != 0;
    const bool strict = (P.flags & DQP_FLAG_STRICT_GET_STEP) != 0;
    if (P.cap) {        // pass 2 of the batch rule: only the flagged QPs, up to the reference's stop
        if (!term_flagged(P, qp)) return;
        maxIter = min(maxIter, P.cap[0]);
    }
    const Ctx C = make_ctx(P, qp);
    const Sh &S = C.S;
    const int N = C.N, M = C.M, E = C.E, NP = C.L.NP, MP = C.L.MP, EP = C.L.EP;
    PROF_T0();
    const Dims D = C.D;
    double *ws = C.ws;
    int status = setup(sm, ws, D, P.Q + qp * P.sQ, P.G + qp * P.sG, P.E > 0 ? P.A + qp * P.sA : nullptr);
    PROF(0);
#ifdef DQP_BIG_PROF
    if (tid == 0) for (int k = 0; k < 8; ++k) sm[S.red + 8 + k] = 0.0;       // potrf's sub-profile: the iterations only
#endif

    // hat-coordinate constants: ph = Lq^-1 p, h, b
    for (int i = tid; i < NP; i += NTHR) sm[S.ph + i] = i < N ? P.p[qp * P.sp + i] : 0.0;
    for (int i = tid; i < MP; i += NTHR) sm[S.hh + i] = i < M ? P.h[qp * P.sh + i] : 0.0;
    for (int i = tid; i < (EP > 0 ? EP : TB); i += NTHR) sm[S.bb + i] = (E > 0 && i < E) ? P.b[qp * P.sb + i] : 0.0;
    __syncthreads();
    trsv_L(sm, S.ph, C.Lq, NP, NP, C.Lqi, S.ct);

    // initial point: d = 1, solve_kkt(p, 0, -h, -b)                                   batch.py:60-74
    for (int i = tid; i < MP; i += NTHR) { sm[S.g + i] = 1.0; sm[S.rz + i] = -sm[S.hh + i]; }
    for (int i = tid; i < (EP > 0 ? EP : TB); i += NTHR) sm[S.ry + i] = -sm[S.bb + i];
    __syncthreads();
    if (!factor_T(sm, ws, D, S.g) && status == DQP_STATUS_OK) status = DQP_STATUS_Q_NOT_PD;
    kkt_wz(sm, ws, D, S.ph, -1, S.rz, S.ry, S.dz);
    kkt_xy(sm, ws, D, S.ph, S.ry, S.dz, S.xh, S.y);
    for (int i = tid; i < MP; i += NTHR) { sm[S.z + i] = i < M ? sm[S.dz + i] : 1.0; sm[S.s + i] = i < M ? -sm[S.dz + i] : 1.0; }
    __syncthreads();
    {   // make s, z >= 1                                                               batch.py:76-86
        double ms = INFINITY, mz = INFINITY;
        for (int i = tid; i < M; i += NTHR) { ms = fmin(ms, sm[S.s + i]); mz = fmin(mz, sm[S.z + i]); }
        ms = wg_min(sm, ms, S.red); mz = wg_min(sm, mz, S.red);
        for (int i = tid; i < M; i += NTHR) {
            if (ms < 0.0) sm[S.s + i] -= ms - 1.0;
            if (mz < 0.0) sm[S.z + i] -= mz - 1.0;
        }
        __syncthreads();
    }
    double *best = C.ws + C.L.oBest, *bxh = best, *bs = best + NP, *bz = bs + MP, *by = bz + MP;
    double bestres = INFINITY;
    bool have_best = false;
    int nNot = 0, iters = 0;

    for (int it = 0; it < maxIter; ++it) {
        // residuals in hat coordinates                                                 batch.py:93-108
        matvecT(sm, S.rxh, C.Gh, NP, M, N, S.z, S.ct);
        if (E > 0) {
            matvecT(sm, S.tn, C.Ah, NP, E, N, S.y, S.ct);
            matvec(sm, S.ry, C.Ah, NP, E, N, S.xh, S.ct);
        }
        matvec(sm, S.rz, C.Gh, NP, M, N, S.xh, S.ct);
        double sz = 0.0, nz2 = 0.0, ny2 = 0.0;
        for (int i = tid; i < NP; i += NTHR) sm[S.rxh + i] = i < N ? sm[S.xh + i] + sm[S.ph + i] + sm[S.rxh + i] + (E > 0 ? sm[S.tn + i] : 0.0) : 0.0;
        for (int i = tid; i < M; i += NTHR) {
            sm[S.rz + i] += sm[S.s + i] - sm[S.hh + i];
            nz2 = fma(sm[S.rz + i], sm[S.rz + i], nz2);
            sz = fma(sm[S.s + i], sm[S.z + i], sz);
        }
        for (int i = tid; i < E; i += NTHR) { sm[S.ry + i] -= sm[S.bb + i]; ny2 = fma(sm[S.ry + i], sm[S.ry + i], ny2); }
        __syncthreads();
        matvec(sm, S.tn2, C.Lq, NP, N, N, S.rxh, S.ct);                    // rx = Lq rxh (Lq has a zero upper part)
        double nx2 = 0.0;
        for (int i = tid; i < N; i += NTHR) nx2 = fma(sm[S.tn2 + i], sm[S.tn2 + i], nx2);
        sz = wg_sum(sm, sz, S.red); nz2 = wg_sum(sm, nz2, S.red); ny2 = wg_sum(sm, ny2, S.red); nx2 = wg_sum(sm, nx2, S.red);
        const double mu = fabs(sz / M);
        const double resid = sqrt(nz2) + sqrt(ny2) + sqrt(nx2) + M * mu;
        PROF(1);
        iters = it + 1;
        if (!have_best || resid < bestres) {                                            // batch.py:119-140
            nNot = 0; have_best = true; bestres = resid;
            for (int i = tid; i < N; i += NTHR) bxh[i] = sm[S.xh + i];
            for (int i = tid; i < M; i += NTHR) { bs[i] = sm[S.s + i]; bz[i] = sm[S.z + i]; }
            for (int i = tid; i < E; i += NTHR) by[i] = sm[S.y + i];
        } else {
            nNot += 1;
        }
        if (batch) {
            if (P.hist && tid == 0) hist_put(P, qp, it, resid, mu);
            if (!(fabs(resid) < INFINITY)) break;
        } else if ((nNot >= P.notImprovedLim && bestres < P.stallTol) || bestres < P.eps || mu > 1e32 ||
                   !(fabs(resid) < INFINITY))
            break;

        for (int i = tid; i < MP; i += NTHR) sm[S.g + i] = i < M ? sm[S.s + i] / sm[S.z + i] : 1.0;    // 1/d, d = z/s
        __syncthreads();
        if (!factor_T(sm, ws, D, S.g) && status == DQP_STATUS_OK) status = DQP_STATUS_Q_NOT_PD;
        PROF(2);
        // affine direction (rs = z => rs/d = s)                                       batch.py:147-163
        kkt_wz(sm, ws, D, S.rxh, S.s, S.rz, S.ry, S.dz);                 // dz_a in S.dz
        PROF(3);
        double ra = INFINITY;
        bool zero_step = false;
        for (int i = tid; i < M; i += NTHR) {
            const double dza = sm[S.dz + i], dsa = (-sm[S.z + i] - dza) * sm[S.g + i];
            sm[S.dsa + i] = dsa;
            if (dza < 0.0) ra = fmin(ra, -sm[S.z + i] / dza);
            if (dsa < 0.0) ra = fmin(ra, -sm[S.s + i] / dsa);
            zero_step |= (dza == 0.0 || dsa == 0.0);
        }
        const double alpha_a = fmin(wg_min(sm, ra, S.red), 1.0);
        double t3 = 0.0;
        for (int i = tid; i < M; i += NTHR) t3 = fma(sm[S.s + i] + alpha_a * sm[S.dsa + i], sm[S.z + i] + alpha_a * sm[S.dz + i], t3);
        t3 = wg_sum(sm, t3, S.red);
        double sig = t3 / sz;
        sig = sig * sig * sig;
        // corrector: rx = rz = ry = 0, rs = (-mu sig + ds_a dz_a)/s;  dz_c = T^-1 (-rs/d)   batch.py:165-181
        for (int i = tid; i < MP; i += NTHR) {
            const double rsc = i < M ? (-mu * sig + sm[S.dsa + i] * sm[S.dz + i]) / sm[S.s + i] : 0.0;
            sm[S.tm + i] = rsc;                                       // kept for ds_c
            sm[S.ds + i] = -rsc * sm[S.g + i];                             // rhs of the corrector z-solve
        }
        __syncthreads();
        trsv_L(sm, S.ds, C.T, MP, MP, C.Ti, S.ct);
        trsv_LT(sm, S.ds, C.T, MP, MP, C.Ti, S.ct);                  // dz_c in S.ds
        PROF(4);
        double rc = INFINITY;
        for (int i = tid; i < M; i += NTHR) {
            const double dzc = sm[S.ds + i], dz = sm[S.dz + i] + dzc, ds = sm[S.dsa + i] + (-sm[S.tm + i] - dzc) * sm[S.g + i];
            sm[S.dz + i] = dz; sm[S.ds + i] = ds;
            if (dz < 0.0) rc = fmin(rc, -sm[S.z + i] / dz);
            if (ds < 0.0) rc = fmin(rc, -sm[S.s + i] / ds);
            zero_step |= (dz == 0.0 || ds == 0.0);
        }
        for (int i = M + tid; i < MP; i += NTHR) { sm[S.dz + i] = 0.0; sm[S.ds + i] = 0.0; }
        __syncthreads();
        const double alpha = fmin(0.999 * wg_min(sm, rc, S.red), 1.0);
        // DQP_FLAG_STRICT_GET_STEP: batch.py:211-214 divides by the step; an exactly-zero component freezes the problem
        if (strict && wg_sum(sm, zero_step ? 1.0 : 0.0, S.red) > 0.0) break;
        kkt_xy(sm, ws, D, S.rxh, S.ry, S.dz, S.dxh, S.te2);
        for (int i = tid; i < N; i += NTHR) sm[S.xh + i] += alpha * sm[S.dxh + i];
        for (int i = tid; i < M; i += NTHR) { sm[S.s + i] += alpha * sm[S.ds + i]; sm[S.z + i] += alpha * sm[S.dz + i]; }
        for (int i = tid; i < E; i += NTHR) sm[S.y + i] += alpha * sm[S.te2 + i];
        __syncthreads();
        PROF(5);
    }
    __syncthreads();
    PROF_OUT();
    if (P.hist && tid == 0) hist_fill(P, qp, iters);
    // back to the caller's coordinates: x = Lq^-T xh
    for (int i = tid; i < NP; i += NTHR) sm[S.tn + i] = i < N ? bxh[i] : 0.0;
    __syncthreads();
    trsv_LT(sm, S.tn, C.Lq, NP, NP, C.Lqi, S.ct);
    for (int i = tid; i < N; i += NTHR) P.zhat[qp * N + i] = sm[S.tn + i];
    for (int i = tid; i < M; i += NTHR) { P.lam[qp * M + i] = bz[i]; P.slack[qp * M + i] = bs[i]; }
    for (int i = tid; i < E; i += NTHR) P.nu[qp * E + i] = by[i];
    if (tid == 0) {
        if (P.info) { P.info[2 * qp] = status; P.info[2 * qp + 1] = iters; }
        if (P.best_resid) P.best_resid[qp] = bestres;
    }
}

// QPFunctionFn.backward (qp.py:128-183) / DenseQPFunction's Solver.backward (qp.py:239-270): one factorisation of
// T = R + D^-1 at the returned iterate, one solve with rhs (dl/dzhat, 0, 0, 0), the six outer-product gradients.
// With DQP_FLAG_BACKWARD_CTX the workspace still holds the forward's Lq, Gh, Ah, L1, GA, R (qp.py:93-95: the
// reference keeps Q_LU, S_LU, R on ctx); without it they are rebuilt.
__global__ __launch_bounds__(NTHR) void backward_kernel(KParams P)
{
    const LP sm = lds_base();
    const long long qp = blockIdx.x;
    const int tid = threadIdx.x;
    const Ctx C = make_ctx(P, qp);
    const Sh &S = C.S;
    const int N = C.N, M = C.M, E = C.E, NP = C.L.NP, MP = C.L.MP, EP = C.L.EP;
    int status = DQP_STATUS_OK;
    const Dims D = C.D;
    double *ws = C.ws;
    if (!(P.flags & DQP_FLAG_BACKWARD_CTX)) status = setup(sm, ws, D, P.Q + qp * P.sQ, P.G + qp * P.sG, P.E > 0 ? P.A + qp * P.sA : nullptr);
    const double *zh = P.zin + qp * N, *lam = P.lamin + qp * M, *slk = P.slackin + qp * M;
    const double *nu = E > 0 ? P.nuin + qp * E : nullptr;
    const bool dense = (P.flags & DQP_FLAG_DENSE_BACKWARD) != 0;
    for (int i = tid; i < MP; i += NTHR)
        sm[S.g + i] = i < M ? (dense ? slk[i] / lam[i] : fmax(slk[i], 1e-8) / fmax(lam[i], 1e-8)) : 1.0;      // qp.py:149
    for (int i = tid; i < NP; i += NTHR) sm[S.rxh + i] = i < N ? P.gin[qp * N + i] : 0.0;
    __syncthreads();
    if (!factor_T(sm, ws, D, S.g) && status == DQP_STATUS_OK) status = DQP_STATUS_Q_NOT_PD;
    trsv_L(sm, S.rxh, C.Lq, NP, NP, C.Lqi, S.ct);
    for (int i = tid; i < MP; i += NTHR) sm[S.rz + i] = 0.0;
    for (int i = tid; i < (EP > 0 ? EP : TB); i += NTHR) sm[S.ry + i] = 0.0;
    __syncthreads();
    kkt_wz(sm, ws, D, S.rxh, -1, S.rz, S.ry, S.dz);                  // dlam
    kkt_xy(sm, ws, D, S.rxh, S.ry, S.dz, S.dxh, S.te2);                   // dxh, dnu
    trsv_LT(sm, S.dxh, C.Lq, NP, NP, C.Lqi, S.ct);                    // dx
    const LP dx = sm + S.dxh, dlam = sm + S.dz, dnu = sm + S.te2;
    for (int i = tid; i < N; i += NTHR) if (P.dp) P.dp[qp * N + i] = dx[i];
    for (int i = tid; i < M; i += NTHR) if (P.dh) P.dh[qp * M + i] = -dlam[i];
    for (int i = tid; i < E; i += NTHR) if (P.db) P.db[qp * E + i] = -dnu[i];
    // outer products: a wavefront per row, lanes along the contiguous nz axis
    const int lane = tid & 63, w = tid >> 6;
    if (P.dQ) {
        double *o = P.dQ + qp * (long long)N * N;
        for (int i = w; i < N; i += 4)
            for (int j = lane; j < N; j += 64) o[(long long)i * N + j] = 0.5 * (dx[i] * zh[j] + zh[i] * dx[j]);
    }
    if (P.dG) {
        double *o = P.dG + qp * (long long)M * N;
        for (int i = w; i < M; i += 4)
            for (int j = lane; j < N; j += 64) o[(long long)i * N + j] = dlam[i] * zh[j] + lam[i] * dx[j];
    }
    if (P.dA && E > 0) {
        double *o = P.dA + qp * (long long)E * N;
        for (int i = w; i < E; i += 4)
            for (int j = lane; j < N; j += 64) o[(long long)i * N + j] = dnu[i] * zh[j] + nu[i] * dx[j];
    }
    if (tid == 0 && P.info) { P.info[2 * qp] = status; P.info[2 * qp + 1] = 0; }
}

template <class K> int launch(K kernel, const KParams &P, void *stream)
{
    const Lay L = layout(P.N, P.M, P.E);
    const size_t lds = lds_doubles(L) * sizeof(double);
    if (lds > 160 * 1024) return DQP_ERR_TOO_LARGE;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DQP_ERR_LAUNCH;
    DQP_LAUNCH(kernel, dim3(P.B), dim3(NTHR), lds, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // namespace big

long long big_workspace_doubles(int N, int M, int E) { return big::layout(N, M, E).total; }
bool big_fits(int N, int M, int E) { return big::lds_doubles(big::layout(N, M, E)) * sizeof(double) <= 160 * 1024; }
int big_forward(const KParams &P, void *stream) { return big::launch(big::forward_kernel, P, stream); }
int big_backward(const KParams &P, void *stream) { return big::launch(big::backward_kernel, P, stream); }

}  // namespace dqp
