// dqp_r16_prims.h -- DPP-row building blocks shared by the 4-QPs-per-wavefront kernels.
//
// Conventions: a QP owns one 16-lane DPP row (r = lane & 15).  "Row-distributed" matrix: row i
// on lane i % 16, slot i / 16, all columns in consecutive VGPRs (double M[S][NC]).
// "Distributed" vector: element i on lane i % 16, slot i / 16 (double v[S]).  Every loop over a
// register index is fully unrolled, so all indices are compile-time constants.
#ifndef DQP_R16_PRIMS_H_
#define DQP_R16_PRIMS_H_
#include <hip/hip_runtime.h>
#include <math.h>

#include "dqp_common.h"

namespace dqp {
namespace r16 {

// ------------------------------------------------------------------ cross-lane primitives
template <int CTRL> __device__ __forceinline__ double dppd(double v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, true);
}

// value of lane k (0..15) of each DPP row, in every lane of that row.  k folds to a constant
// after unrolling, leaving a single v_mov_b64_dpp row_newbcast:k.
__device__ __forceinline__ double rb(double v, int k)
{
    switch (k & 15) {
    case 0: return dppd<0x150>(v);   case 1: return dppd<0x151>(v);
    case 2: return dppd<0x152>(v);   case 3: return dppd<0x153>(v);
    case 4: return dppd<0x154>(v);   case 5: return dppd<0x155>(v);
    case 6: return dppd<0x156>(v);   case 7: return dppd<0x157>(v);
    case 8: return dppd<0x158>(v);   case 9: return dppd<0x159>(v);
    case 10: return dppd<0x15a>(v);  case 11: return dppd<0x15b>(v);
    case 12: return dppd<0x15c>(v);  case 13: return dppd<0x15d>(v);
    case 14: return dppd<0x15e>(v);  default: return dppd<0x15f>(v);
    }
}
#define BC(vec, k) rb((vec)[(k) >> 4], (k) & 15)   /* element k of a distributed vector */

__device__ __forceinline__ double row_sum(double v)
{
    v += dppd<0x128>(v); v += dppd<0x124>(v); v += dppd<0x122>(v); v += dppd<0x121>(v);
    return v;
}
__device__ __forceinline__ double row_min(double v)
{
    v = fmin(v, dppd<0x128>(v)); v = fmin(v, dppd<0x124>(v));
    v = fmin(v, dppd<0x122>(v)); v = fmin(v, dppd<0x121>(v));
    return v;
}

__device__ __forceinline__ double row_max(double v)
{
    v = fmax(v, dppd<0x128>(v)); v = fmax(v, dppd<0x124>(v));
    v = fmax(v, dppd<0x122>(v)); v = fmax(v, dppd<0x121>(v));
    return v;
}

// Pins a value's definition where it is written: without it the compiler may sink a per-step
// `if (r == k) rd = x` select to the first use of rd, keeping all N candidates x alive (60 VGPRs
// through the next phase) -- seen in the null-space backward kernel: 5.7 KB/lane of spills.
#define PIN(x) asm volatile("" : "+v"(x))

// keep ? v : (v with its high word cleared).  The cleared value is 0 or a positive denormal
// below 2^-1042, which every FMA on normal-range data absorbs exactly -- one v_cndmask instead of
// the two a full 64-bit select costs (these kernels issue one instruction per 4 cycles, so
// predication selects were ~20 % of the PDIPM iteration).
__device__ __forceinline__ double mask_hi(double v, bool keep)
{
    return __hiloint2double(keep ? __double2hiint(v) : 0, __double2loint(v));
}

// full-precision reciprocal / reciprocal square root from the hardware estimates + 2 Newton steps
__device__ __forceinline__ double frcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double frsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    y = y * fma(-0.5 * d * y, y, 1.5);
    y = y * fma(-0.5 * d * y, y, 1.5);
    return y;
}

// Diagnostic phase stamps: compiled in only with -DDQP_STAMPS (tools/stamps.py builds that
// variant itself).  Even a never-taken stamp branch perturbs register allocation of the
// fully unrolled kernels by 2-3x, so the shipped build contains none.
#if defined(DQP_MARKS)   /* asm comments at the phase boundaries, for reading the .s */
#define DQP_STR2(x) #x
#define DQP_STR(x) DQP_STR2(x)
#define STAMP(P, i) asm volatile("; DQPMARK " DQP_STR(i))
#elif defined(DQP_SETUP_STOP)   /* measurement builds (tools/setup_phases.py): the wavefront ends at phase boundary DQP_SETUP_STOP */
#define STAMP(P, i)                                                                  \
    do {                                                                             \
        if ((i) == DQP_SETUP_STOP) {                                                 \
            const double a__ = stop_sum<DQP_SETUP_STOP>(st);                         \
            (P).zhat[qp * 2 + (r & 1)] = a__;                                        \
            __builtin_amdgcn_endpgm();                                               \
        }                                                                            \
    } while (0)
#elif !defined(DQP_STAMPS)
#define STAMP(P, i) do { } while (0)
#else
#define STAMP(P, i)                                                                  \
    do {                                                                             \
        if ((P).stamps) {                                                            \
            __builtin_amdgcn_sched_barrier(0);                                       \
            const unsigned long long t__ = __builtin_readcyclecounter();             \
            if (threadIdx.x == 0) (P).stamps[blockIdx.x * 16 + (i)] = t__;           \
            __builtin_amdgcn_sched_barrier(0);                                       \
        }                                                                            \
    } while (0)
#endif

#if defined(DQP_STAMPS_C)     /* experiment: the 16 slots record the reflector loop of the null-space setup */
#undef STAMP
#define STAMP(P, i) do { } while (0)
#define STAMPC(P, i)                                                                 \
    do {                                                                             \
        if ((P).stamps) {                                                            \
            __builtin_amdgcn_sched_barrier(0);                                       \
            const unsigned long long t__ = __builtin_readcyclecounter();             \
            if (threadIdx.x == 0) (P).stamps[blockIdx.x * 16 + (i)] = t__;           \
            __builtin_amdgcn_sched_barrier(0);                                       \
        }                                                                            \
    } while (0)
#endif

constexpr __host__ __device__ int slots(int n) { return (n + 15) / 16; }
constexpr __host__ __device__ int tri(int i) { return i * (i + 1) / 2; }

// ------------------------------------------------------------------ mat-vec building blocks
// y[s] (+)= sum_j M[s][j] * x_j          (M row-distributed SR x NC, x distributed length NC)
template <int SR, int NC, int SX>
__device__ __forceinline__ void mv_nat(const double (&Mx)[SR][NC], const double (&x)[SX],
                                       double (&y)[SR], bool accumulate)
{
    if (!accumulate) {
#pragma unroll
        for (int s = 0; s < SR; ++s) y[s] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const double xb = BC(x, j);
#pragma unroll
        for (int s = 0; s < SR; ++s) y[s] = fma(Mx[s][j], xb, y[s]);
    }
}

// Reduce-scatter over the 16 lanes of a DPP row: on entry every lane holds its partials v[k]
// of 16 column sums; the return value on lane k is the total of column k.  Four mirror
// butterflies (row_mirror, row_half_mirror, quad reverse, quad swap) halve the live values.
__device__ __forceinline__ double reduce_scatter16(const double (&v)[16], int r)
{
    const bool h8 = (r & 8) != 0, h4 = (r & 4) != 0, h2 = (r & 2) != 0, h1 = (r & 1) != 0;
    double a[8], b4[4], c2[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double keep = h8 ? v[k + 8] : v[k], send = h8 ? v[k] : v[k + 8];
        a[k] = keep + dppd<0x140>(send);                 // row_mirror: l <-> 15-l
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double keep = h4 ? a[k + 4] : a[k], send = h4 ? a[k] : a[k + 4];
        b4[k] = keep + dppd<0x141>(send);                // row_half_mirror: l <-> 7-l
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double keep = h2 ? b4[k + 2] : b4[k], send = h2 ? b4[k] : b4[k + 2];
        c2[k] = keep + dppd<0x1b>(send);                 // quad_perm [3,2,1,0]
    }
    const double keep = h1 ? c2[1] : c2[0], send = h1 ? c2[0] : c2[1];
    return keep + dppd<0xb1>(send);                      // quad_perm [1,0,3,2]
}

// y = M^T v   (M row-distributed SR x NC, v distributed over the rows; y distributed length NC)
template <int SR, int NC, int SY>
__device__ __forceinline__ void mv_tr(const double (&Mx)[SR][NC], const double (&v)[SR],
                                      double (&y)[SY], int r)
{
#pragma unroll
    for (int g = 0; g < SY; ++g) {
        double p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = 16 * g + k;
            if (c < NC) {
                double a = Mx[0][c < NC ? c : 0] * v[0];
#pragma unroll
                for (int s = 1; s < SR; ++s) a = fma(Mx[s][c < NC ? c : 0], v[s], a);
                p[k] = a;
            } else {
                p[k] = 0.0;
            }
        }
        y[g] = reduce_scatter16(p, r);
    }
}

// ------------------------------------------------------------------ factorizations in registers
// Lower Cholesky of the row-distributed SPD matrix (in place; strict upper part zeroed).
// rd[s] = 1/L[i][i] for the lane's rows.  Returns false on a pivot that is not positive, or --
// with reltol > 0 -- that has collapsed below reltol x the largest pivot seen (a numerically
// singular matrix: duplicated / dependent equality rows give a round-off-sized positive pivot).
template <int S, int N>
__device__ __forceinline__ bool chol_rows(double (&L)[S][N], double (&rd)[S], int r, double reltol = 0.0)
{
    bool ok = true;
    double pmax = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) rd[s] = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int sk = k >> 4, lk = k & 15;
        double dk = rb(L[sk][k], lk);
        if (!(dk > reltol * pmax)) { ok = false; dk = 1.0; }
        pmax = fmax(pmax, dk);
        const double ri = frsqrt(dk);
        if (r == lk) rd[sk] = ri;
        PIN(rd[sk]);
        double col[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (16 * s + 15 < k) col[s] = 0.0;                              // rows all above k
            else if (16 * s >= k) col[s] = L[s][k] * ri;                    // rows all >= k
            else col[s] = (r >= lk) ? L[s][k] * ri : 0.0;
            L[s][k] = col[s];
        }
#pragma unroll
        for (int j = k + 1; j < N; ++j) {
            const double cj = BC(col, j);
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (16 * s + 15 >= j) L[s][j] = fma(-col[s], cj, L[s][j]);   // only rows i >= j matter
        }
    }
    return ok;
}

// Unpivoted LU of the row-distributed matrix (in place: unit-lower multipliers below the
// diagonal, U on/above).  rdu[s] = 1/U[i][i].
template <int S, int N>
__device__ __forceinline__ void lu_rows(double (&T)[S][N], double (&rdu)[S], int r)
{
#pragma unroll
    for (int s = 0; s < S; ++s) rdu[s] = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int sk = k >> 4, lk = k & 15;
        const double rp = frcp(rb(T[sk][k], lk));
        if (r == lk) rdu[sk] = rp;
        PIN(rdu[sk]);
        double l[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (16 * s + 15 <= k) l[s] = 0.0;
            else if (16 * s > k) { l[s] = T[s][k] * rp; T[s][k] = l[s]; }
            else { const bool a = r > lk; l[s] = mask_hi(T[s][k], a) * rp; T[s][k] = a ? l[s] : T[s][k]; }
        }
#pragma unroll
        for (int j = k + 1; j < N; ++j) {
            const double ub = rb(T[sk][j], lk);
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (16 * s + 15 > k) T[s][j] = fma(-l[s], ub, T[s][j]);
        }
    }
}

// b <- T^-1 b with the LU above.  In the U sweep a lane keeps its own y_k unscaled (the pivot
// scaling is applied to the broadcast copy and, once, to the whole vector at the end), so the
// only per-step predication is the one-instruction triangle mask.
template <int S, int N>
__device__ __forceinline__ void lu_solve(const double (&T)[S][N], const double (&rdu)[S],
                                         double (&b)[S], int r)
{
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {                  // L y = b
        const int lk = k & 15;
        const double bk = BC(b, k);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (16 * s + 15 <= k) continue;
            if (16 * s > k) b[s] = fma(-T[s][k], bk, b[s]);
            else b[s] = fma(-mask_hi(T[s][k], r > lk), bk, b[s]);
        }
    }
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {                 // U x = y
        const int sk = k >> 4, lk = k & 15;
        const double xk = rb(b[sk] * rdu[sk], lk);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (16 * s > k) continue;
            if (16 * s + 15 < k) b[s] = fma(-T[s][k], xk, b[s]);
            else b[s] = fma(-mask_hi(T[s][k], r < lk), xk, b[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) b[s] *= rdu[s];
}

// b <- L^-1 b, L lower triangular row-distributed in registers (rd = reciprocal diagonal)
template <int S, int N>
__device__ __forceinline__ void trsv_rows(const double (&L)[S][N], const double (&rd)[S],
                                          double (&b)[S], int r)
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int sk = k >> 4, lk = k & 15;
        const double yk = rb(b[sk] * rd[sk], lk);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (16 * s + 15 < k) continue;
            if (16 * s > k) b[s] = fma(-L[s][k], yk, b[s]);
            else b[s] = (r == lk) ? yk : fma(r > lk ? -L[s][k] : 0.0, yk, b[s]);
        }
    }
}

// ------------------------------------------------------------------ packed-triangle LDS helpers
// LDS holds, per QP, the packed lower triangle P[tri(i) + j], j <= i.
template <int S, int N>
__device__ __forceinline__ void tri_store(double *P, const double (&L)[S][N], int r, double *dummy)
{
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = r + 16 * s;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j > 16 * s + 15) continue;
            double *dst = (i < N && j <= i) ? P + tri(i) + j : dummy;
            *dst = L[s][j];
        }
    }
}

// y = L x with L the packed lower triangle in LDS (x, y distributed length N)
template <int S, int N>
__device__ __forceinline__ void tri_mv(const double *P, const double (&x)[S], double (&y)[S], int r)
{
#pragma unroll
    for (int s = 0; s < S; ++s) y[s] = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double xb = BC(x, j);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (16 * s + 15 < j) continue;
            const int i = r + 16 * s;
            const int ic = i < N ? i : N - 1;
            const double v = P[tri(ic) + (j <= ic ? j : 0)];
            y[s] = fma(mask_hi(v, j <= i && i < N), xb, y[s]);
        }
    }
}

// b <- L^-1 b with L packed in LDS (forward substitution; rd distributed reciprocal diagonal)
template <int S, int N>
__device__ __forceinline__ void tri_solve(const double *P, const double (&rd)[S], double (&b)[S], int r)
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int sk = k >> 4, lk = k & 15;
        const double yk = rb(b[sk] * rd[sk], lk);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (16 * s + 15 < k) continue;
            const int i = r + 16 * s;
            const int ic = i < N ? i : N - 1;
            const double v = P[tri(ic) + (k <= ic ? k : 0)];
            if (16 * s > k) b[s] = fma((i < N) ? -v : 0.0, yk, b[s]);
            else b[s] = (r == lk) ? yk : fma((r > lk && i < N) ? -v : 0.0, yk, b[s]);
        }
    }
}

// b <- L^-T b with L packed in LDS: x_j = (b_j - sum_{i>j} L[i][j] x_i) / L[j][j]; the sum over
// rows is a row reduction (the transposed access pattern of row-distributed storage).
template <int S, int N>
__device__ __forceinline__ void tri_solve_T(const double *P, const double (&rd)[S], double (&b)[S], int r)
{
#pragma unroll
    for (int j = N - 1; j >= 0; --j) {
        const int sj = j >> 4, lj = j & 15;
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (16 * s + 15 <= j) continue;
            const int i = r + 16 * s;
            const int ic = i < N ? i : N - 1;
            const double v = P[tri(ic) + (j <= ic ? j : 0)];
            part = fma((i > j && i < N) ? v : 0.0, b[s], part);
        }
        const double tot = row_sum(part);
        if (r == lj) b[sj] = (b[sj] - tot) * rd[sj];
    }
}

// distributed vector <-> LDS (each lane touches only its own elements: no barrier needed)
template <int S>
__device__ __forceinline__ void vec_put(double *P, const double (&v)[S], int n, int r, double *dummy)
{
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = r + 16 * s;
        double *dst = i < n ? P + i : dummy;
        *dst = v[s];
    }
}
template <int S>
__device__ __forceinline__ void vec_get(const double *P, double (&v)[S], int n, int r)
{
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = r + 16 * s;
        const double x = P[i < n ? i : n - 1];
        v[s] = i < n ? x : 0.0;
    }
}

// Row-distributed load of an nrows x NC matrix (rows beyond nrows are zero).
// The same row-distributed load staged through LDS: the 16 lanes of a QP copy the contiguous matrix
// with 16 bytes per lane -- 256 contiguous bytes per load instruction and QP, instead of 16 rows 8 NC
// bytes apart (64 distinct cache lines per instruction over the four QPs of a wavefront) -- one register
// slot (16 rows) at a time, and each lane reads its row back from LDS.  `stage` needs 16 NC doubles.
// V2: the matrix starts on a 16-byte boundary for every QP (rows x NC even).
template <int S, int NC, bool V2>
__device__ __forceinline__ void load_rows_staged(const double *src, int nrows, double (&dst)[S][NC], int r, double *stage)
{
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int nr = nrows - 16 * s < 16 ? nrows - 16 * s : 16;       // rows of this slot
        const double *blk = src + 16 * s * NC;
        if (V2 && (reinterpret_cast<unsigned long long>(blk) & 15ull) == 0) {      // (a view may start on an 8-byte boundary)
            const double2 *b2 = reinterpret_cast<const double2 *>(blk);
            double2 *s2 = reinterpret_cast<double2 *>(stage);
#pragma unroll
            for (int k = 0; k < (16 * NC / 2 + 15) / 16; ++k) {
                const int e = r + 16 * k;
                if (e < nr * NC / 2) s2[e] = b2[e];
            }
            if ((nr * NC) & 1) { if (r == 0) stage[nr * NC - 1] = blk[nr * NC - 1]; }
        } else {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int e = r + 16 * k;
                if (e < nr * NC) stage[e] = blk[e];
            }
        }
        __syncthreads();
        const double *row = stage + (r < nr ? r : 0) * NC;
#pragma unroll
        for (int j = 0; j < NC; ++j) dst[s][j] = row[j];
        if (16 * s + 15 >= nrows) {
            const double keep = r < nr ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) dst[s][j] *= keep;
        }
        __syncthreads();
    }
}

template <int S, int NC>
__device__ __forceinline__ void load_rows(const double *src, int nrows, double (&dst)[S][NC], int r)
{
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = r + 16 * s;
        const double *row = src + (i < nrows ? i : nrows - 1) * NC;
        const double keep = i < nrows ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < NC; ++j) dst[s][j] = row[j];
        if (16 * s + 15 >= nrows) {
#pragma unroll
            for (int j = 0; j < NC; ++j) dst[s][j] *= keep;
        }
    }
}

// ------------------------------------------------------------------ Schur complement factor
// (C: a kernel Cfg with M, SM, oR, oDummy)
// T = R + diag(dinv) (full square, row-distributed) from the packed triangle in LDS, then LU.
// Each lane first rewrites its own diagonal entries of the LDS triangle with R_ii + dinv_i (only
// the owning lane ever reads them back), so the load needs no per-element diagonal select.
template <class C>
__device__ __forceinline__ void factor_T(double *lds, double (&T)[C::SM][C::M],
                                         const double (&rdiag)[C::SM], const double (&dinv)[C::SM],
                                         double (&rdu)[C::SM], int r)
{
    constexpr int M = C::M, SM = C::SM;
    double *Rp = lds + C::oR;
#pragma unroll
    for (int s = 0; s < SM; ++s) {
        const int i = r + 16 * s;
        double *dst = i < M ? Rp + tri(i) + i : lds + C::oDummy + r;
        *dst = rdiag[s] + dinv[s];
    }
#pragma unroll
    for (int s = 0; s < SM; ++s) {
        const int i = r + 16 * s;
        const int ic = i < M ? i : M - 1;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            int off;
            if (16 * s > j) off = tri(ic) + j;                          // all rows below column j
            else if (16 * s + 15 < j) off = tri(j) + ic;                // all rows above
            else off = (j <= ic) ? tri(ic) + j : tri(j) + ic;
            const double v = Rp[off];
            T[s][j] = (16 * s + 15 >= M) ? mask_hi(v, i < M) : v;       // pad rows: ~zero rows
        }
    }
    lu_rows<SM, M>(T, rdu, r);
}

}  // namespace r16
}  // namespace dqp
#endif
