// dqp_pdipm.hip -- fused PDIPM forward + KKT backward for small dense QPs on gfx950.
//
// One QP per 64-lane wavefront (one single-wave workgroup per QP).  Every matrix the
// interior-point iteration touches lives in LDS for the whole solve; every vector lives in
// registers, element i on lane i; the only HBM traffic is the coalesced read of
// (Q,p,G,h,A,b) and the write of (zhat,lam,nu,slack).  There is no host synchronisation and
// no per-iteration kernel launch (the reference does ~60 launches + 2 host syncs per
// iteration: qpth/solvers/pdipm/batch.py:91-204).
//
// Math.  The reference eliminates the KKT system with LU(Q) and a block LU of
//   S = [A Q^-1 A^T, A Q^-1 G^T; G Q^-1 A^T, G Q^-1 G^T + D^-1]      (batch.py:351-428).
// Newton's method is affine invariant, so we run the SAME iteration in the coordinates
//   xh = Lq^T x  (Q = Lq Lq^T),  yt = L1^T y  (A Q^-1 A^T = L1 L1^T)
// where Q becomes I and A becomes At = L1^-1 A Lq^-T with orthonormal rows:
//   Gh = G Lq^-T,  R = Gh (I - At^T At) Gh^T  (== the reference's Schur complement R),
//   T  = R + diag(s/z)  factored per iteration as L D L^T (the symmetric form of the
//        unpivoted LU the reference uses on GPUs, batch.py:8-19).
// A KKT solve is then 4 small mat-vecs with Gh/At plus one LDL^T solve; s, z (lam) and the
// step-length rule are untouched by the change of variables, residual norms are mapped back
// (||rx|| = ||Lq rxh||, ||ry|| = ||L1 ryt||) so best-iterate selection follows batch.py:98-140.
//
// Reference functions covered: a2 pre_factor_kkt, a3 forward, a4 factor_kkt,
// a5 solve_kkt, a6 get_step, a8 QPFunctionFn.backward, a9/a10 DenseQPFunction (same Newton
// systems; flag selects its un-clamped backward).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "dqp_common.h"
#include "dqp_dyn_models.h"

using namespace dqp;

namespace {

#define WSYNC() __syncthreads()

__device__ __forceinline__ double bcast(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

template <int CTRL> __device__ __forceinline__ double dppd(double v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, true);
}

// all-reduce inside each 16-lane DPP row (row_ror 8/4/2/1), then combine the 4 rows
__device__ __forceinline__ double wave_sum(double v)
{
    v += dppd<0x128>(v); v += dppd<0x124>(v); v += dppd<0x122>(v); v += dppd<0x121>(v);
    return (bcast(v, 0) + bcast(v, 16)) + (bcast(v, 32) + bcast(v, 48));
}

__device__ __forceinline__ double wave_min(double v)
{
    v = fmin(v, dppd<0x128>(v)); v = fmin(v, dppd<0x124>(v));
    v = fmin(v, dppd<0x122>(v)); v = fmin(v, dppd<0x121>(v));
    return fmin(fmin(bcast(v, 0), bcast(v, 16)), fmin(bcast(v, 32), bcast(v, 48)));
}

// The mat-vecs are unrolled over a compile-time size class C (16/32/64 >= every dimension) in
// chunks of 8 so that the LDS reads of a chunk are all in flight before the first FMA; indices
// past the runtime size are clamped (uniform, SALU) and their products masked.
constexpr int CH = 8;

// y_i = sum_{j<n} M[i][j] x_j for lanes i < m (0 elsewhere).  x: element j on lane j.
template <int C>
__device__ __forceinline__ double matvec(const double *Mx, int ld, int m, int n, double x, int lane)
{
    const double *row = Mx + (lane < m ? lane : 0) * ld;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int jb = 0; jb < C; jb += CH) {
        if (jb < n) {
            double v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = row[min(jb + u, n - 1)];
#pragma unroll
            for (int u = 0; u < CH; u += 2) {
                a0 = fma(v[u], (jb + u < n) ? bcast(x, jb + u) : 0.0, a0);
                a1 = fma(v[u + 1], (jb + u + 1 < n) ? bcast(x, jb + u + 1) : 0.0, a1);
            }
        }
    }
    return lane < m ? a0 + a1 : 0.0;
}

// y_j = sum_{i<m} M[i][j] x_i for lanes j < n (0 elsewhere).  x: element i on lane i.
template <int C>
__device__ __forceinline__ double matvecT(const double *Mx, int ld, int m, int n, double x, int lane)
{
    const double *col = Mx + (lane < n ? lane : 0);
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int ib = 0; ib < C; ib += CH) {
        if (ib < m) {
            double v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = col[min(ib + u, m - 1) * ld];
#pragma unroll
            for (int u = 0; u < CH; u += 2) {
                a0 = fma(v[u], (ib + u < m) ? bcast(x, ib + u) : 0.0, a0);
                a1 = fma(v[u + 1], (ib + u + 1 < m) ? bcast(x, ib + u + 1) : 0.0, a1);
            }
        }
    }
    return lane < n ? a0 + a1 : 0.0;
}

// y_i = sum_{j<=i} L[i][j] x_j (lower-triangular mat-vec; the strict upper part is masked)
template <int C>
__device__ __forceinline__ double trimatvec(const double *L, int ld, int n, double x, int lane)
{
    const double *row = L + (lane < n ? lane : 0) * ld;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int jb = 0; jb < C; jb += CH) {
        if (jb < n) {
            double v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = row[min(jb + u, n - 1)];
#pragma unroll
            for (int u = 0; u < CH; u += 2) {
                const double x0 = (jb + u < n) ? bcast(x, jb + u) : 0.0;
                const double x1 = (jb + u + 1 < n) ? bcast(x, jb + u + 1) : 0.0;
                a0 = fma(jb + u <= lane ? v[u] : 0.0, x0, a0);
                a1 = fma(jb + u + 1 <= lane ? v[u + 1] : 0.0, x1, a1);
            }
        }
    }
    return lane < n ? a0 + a1 : 0.0;
}

// In-place lower Cholesky of the n x n matrix in LDS; rd = 1/L[i][i] on lane i.
__device__ __forceinline__ bool chol_factor(double *L, int ld, int n, int lane, double &rd, double reltol = 0.0)
{
    bool ok = true;
    rd = 0.0;
    double pmax = 0.0;      // reltol > 0: a pivot collapsed below reltol x the largest one = singular
    for (int k = 0; k < n; ++k) {
        WSYNC();
        double dk = L[k * ld + k];
        if (!(dk > reltol * pmax)) { ok = false; dk = 1.0; }
        pmax = fmax(pmax, dk);
        const double sq = sqrt(dk), r = 1.0 / sq;
        const bool act = lane > k && lane < n;
        double c = 0.0;
        if (lane == k) { rd = r; L[k * ld + k] = sq; }
        if (act) { c = L[lane * ld + k] * r; L[lane * ld + k] = c; }
        for (int i = k + 1; i < n; ++i) {
            const double ci = bcast(c, i);
            if (act && lane <= i) L[i * ld + lane] = fma(-ci, c, L[i * ld + lane]);
        }
    }
    WSYNC();
    return ok;
}

// In-place L D L^T of the symmetric n x n matrix in LDS (unit L in the strict lower part).
// Returns 1/D[i] on lane i.
__device__ double ldl_factor(double *T, int ld, int n, int lane)
{
    double rdiag = 0.0;
    for (int k = 0; k < n; ++k) {
        WSYNC();
        const double r = 1.0 / T[k * ld + k];
        const bool act = lane > k && lane < n;
        double c = 0.0, l = 0.0;
        if (lane == k) rdiag = r;
        if (act) { c = T[lane * ld + k]; l = c * r; T[lane * ld + k] = l; }
        for (int i = k + 1; i < n; ++i) {
            const double li = bcast(l, i);
            if (act && lane <= i) T[i * ld + lane] = fma(-li, c, T[i * ld + lane]);
        }
    }
    WSYNC();
    return rdiag;
}

__device__ __forceinline__ double ldl_solve(const double *T, int ld, int n, double b, double rdiag, int lane)
{
    for (int k = 0; k < n - 1; ++k) {
        const double bk = bcast(b, k);
        if (lane > k && lane < n) b = fma(-T[lane * ld + k], bk, b);
    }
    b *= rdiag;
    for (int k = n - 1; k > 0; --k) {
        const double bk = bcast(b, k);
        if (lane < k) b = fma(-T[k * ld + lane], bk, b);
    }
    return b;
}


// ---- register-resident T (per-iteration Schur complement) ---------------------------------
// Lane j holds column j (== row j) of the symmetric matrix in t[0..MAXM).  After ldl_reg(),
// t[k] for k < lane is L[lane][k] (row of the unit-lower factor), t[i] for i > lane is
// L[i][lane] (its column), and the return value is 1/D[lane].  All indices are compile-time
// after unrolling, so t[] stays in VGPRs; the only cross-lane traffic is v_readlane.
template <int MAXM>
__device__ __forceinline__ double ldl_reg(double (&t)[MAXM], int n, int lane)
{
    double rdiag = 0.0;
#pragma unroll
    for (int k = 0; k < MAXM; ++k) {
        if (k < n) {
            const double r = 1.0 / bcast(t[k], k);
            if (lane == k) rdiag = r;
            const bool act = lane > k;
            const double c = act ? t[k] : 0.0;   // D_k L[lane][k]; 0 keeps finished lanes intact
            const double l = c * r;              // L[lane][k]
            t[k] = act ? l : t[k];
#pragma unroll
            for (int i = k + 1; i < MAXM; ++i)
                if (i < n) t[i] = fma(-bcast(l, i), c, t[i]);
        }
    }
    // lane j still holds D_j L[i][j] in t[i], i > j (its column as of step j): scale to L[i][j]
#pragma unroll
    for (int i = 1; i < MAXM; ++i) t[i] = (i > lane) ? t[i] * rdiag : t[i];
    return rdiag;
}

template <int MAXM>
__device__ __forceinline__ double ldl_reg_solve(const double (&t)[MAXM], int n, double b,
                                                double rdiag, int lane)
{
#pragma unroll
    for (int k = 0; k < MAXM - 1; ++k)
        if (k < n - 1) {
            const double bk = bcast(b, k);
            if (lane > k) b = fma(-t[k], bk, b);
        }
    b *= rdiag;
#pragma unroll
    for (int k = MAXM - 1; k > 0; --k)
        if (k < n) {
            const double bk = bcast(b, k);
            if (lane < k) b = fma(-t[k], bk, b);
        }
    return lane < n ? b : 0.0;
}

// L y = b (L lower, non-unit; rd = reciprocal diagonal, element i on lane i)
__device__ __forceinline__ double trsv_L(const double *L, int ld, int n, double b, double rd, int lane)
{
    for (int k = 0; k < n; ++k) {
        const double yk = bcast(b * rd, k);
        if (lane == k) b = yk;
        else if (lane > k && lane < n) b = fma(-L[lane * ld + k], yk, b);
    }
    return b;
}

// L^T x = b
__device__ __forceinline__ double trsv_LT(const double *L, int ld, int n, double b, double rd, int lane)
{
    for (int k = n - 1; k >= 0; --k) {
        const double xk = bcast(b * rd, k);
        if (lane == k) b = xk;
        else if (lane < k) b = fma(-L[k * ld + lane], xk, b);
    }
    return b;
}

struct Lds {
    double *Lq, *Gh, *At, *L1, *R, *T, *W;
};

__device__ __forceinline__ Lds carve(double *sm, const KParams &P)
{
    Lds s;
    s.Lq = sm;
    s.Gh = s.Lq + P.N * P.ldz;
    s.At = s.Gh + P.M * P.ldz;
    s.L1 = s.At + P.E * P.ldz;
    s.R = s.L1 + P.E * P.lde;
    s.W = s.R;   // W = Gh At^T is dead before R is written (qp_setup), so they share storage
    s.T = s.R + P.M * (P.ldm > P.lde ? P.ldm : P.lde);
    return s;
}

// One-time factorisations (reference: pre_factor_kkt, batch.py:377-428).
// On return: Lq (chol Q), Gh = G Lq^-T, At = L1^-1 A Lq^-T, L1 = chol(A Q^-1 A^T),
// R = Gh (I - At^T At) Gh^T; rdq / rd1 = reciprocal diagonals of Lq / L1 on lane i.
__device__ __forceinline__ int qp_setup(const KParams &P, const Lds &S, int qp, int lane, double &rdq, double &rd1)
{
    const int N = P.N, M = P.M, E = P.E, ldz = P.ldz, ldm = P.ldm, lde = P.lde, ldt = P.ldt;
    const double *Q = P.Q + (long long)qp * P.sQ;
    const double *G = P.G + (long long)qp * P.sG;
    const double *A = E ? P.A + (long long)qp * P.sA : nullptr;
    int status = DQP_STATUS_OK;

    // coalesced row loads into padded LDS rows
    for (int i = 0; i < N; ++i)
        if (lane < N) S.Lq[i * ldz + lane] = Q[i * N + lane];
    for (int i = 0; i < M; ++i)
        if (lane < N) S.Gh[i * ldz + lane] = G[i * N + lane];
    for (int i = 0; i < E; ++i)
        if (lane < N) S.At[i * ldz + lane] = A[i * N + lane];

    if (!chol_factor(S.Lq, ldz, N, lane, rdq)) status = DQP_STATUS_Q_NOT_PD;

    // rows of G and A -> solve Lq r = row  (lanes = right-hand sides)
    for (int base = 0; base < M + E; base += WAVE) {
        const int r = base + lane;
        if (r < M + E) {
            double *row = r < M ? S.Gh + r * ldz : S.At + (r - M) * ldz;
            for (int k = 0; k < N; ++k) {
                double v = row[k];
                const double *Lk = S.Lq + k * ldz;
                for (int j = 0; j < k; ++j) v = fma(-Lk[j], row[j], v);
                row[k] = v / Lk[k];
            }
        }
    }
    WSYNC();

    rd1 = 0.0;
    if (E > 0) {
        // S11 = Ah Ah^T
        for (int base = 0; base < E * E; base += WAVE) {
            const int idx = base + lane;
            if (idx < E * E) {
                const int i = idx / E, j = idx - i * E;
                const double *ri = S.At + i * ldz, *rj = S.At + j * ldz;
                double a = 0.0;
                for (int k = 0; k < N; ++k) a = fma(ri[k], rj[k], a);
                S.L1[i * lde + j] = a;
            }
        }
        if (!chol_factor(S.L1, lde, E, lane, rd1, 1e-13) && status == DQP_STATUS_OK)
            status = DQP_STATUS_A_RANK_DEF;
        // At = L1^-1 Ah  (lanes = columns)
        if (lane < N) {
            for (int i = 0; i < E; ++i) {
                double v = S.At[i * ldz + lane];
                const double *Li = S.L1 + i * lde;
                for (int j = 0; j < i; ++j) v = fma(-Li[j], S.At[j * ldz + lane], v);
                S.At[i * ldz + lane] = v / Li[i];
            }
        }
        WSYNC();
        // W = Gh At^T (M x E)
        for (int base = 0; base < M * E; base += WAVE) {
            const int idx = base + lane;
            if (idx < M * E) {
                const int i = idx / E, e = idx - i * E;
                const double *gi = S.Gh + i * ldz, *ae = S.At + e * ldz;
                double a = 0.0;
                for (int k = 0; k < N; ++k) a = fma(gi[k], ae[k], a);
                S.W[i * lde + e] = a;
            }
        }
        WSYNC();
    }
    // Gbar = Gh - W At  (staged in the T buffer, row stride ldt), then R = Gbar Gbar^T
    if (lane < N) {
        for (int i = 0; i < M; ++i) {
            double v = S.Gh[i * ldz + lane];
            for (int e = 0; e < E; ++e) v = fma(-S.W[i * lde + e], S.At[e * ldz + lane], v);
            S.T[i * ldt + lane] = v;
        }
    }
    WSYNC();
    for (int base = 0; base < M * M; base += WAVE) {
        const int idx = base + lane;
        if (idx < M * M) {
            const int i = idx / M, j = idx - i * M;
            const double *gi = S.T + i * ldt, *gj = S.T + j * ldt;
            double a = 0.0;
            for (int k = 0; k < N; ++k) a = fma(gi[k], gj[k], a);
            S.R[i * ldm + j] = a;
        }
    }
    WSYNC();
    return status;
}

// T = R + diag(dinv) loaded column-per-lane into registers, then L D L^T.  dinv: element i
// on lane i.  Returns 1/D on lane i.
template <int MAXM>
__device__ __forceinline__ double factor_T(const KParams &P, const Lds &S, double (&t)[MAXM],
                                           double dinv, int lane)
{
    const int M = P.M, ldm = P.ldm;
    const double *col = S.R + (lane < M ? lane : 0);   // R symmetric: column == row
#pragma unroll
    for (int i = 0; i < MAXM; ++i) t[i] = (i < M) ? col[i * ldm] : 0.0;
#pragma unroll
    for (int i = 0; i < MAXM; ++i)
        if (i < M && lane == i) t[i] += dinv;
    return ldl_reg<MAXM>(t, M, lane);
}

// z-part of solve_kkt in hat coordinates: returns wz = dz.
//   rxh (lane<N), rsd = rs/d (lane<M), rz (lane<M), ryt (lane<E)
template <int MAXM>
__device__ __forceinline__ double kkt_wz(const KParams &P, const Lds &S, const double (&t)[MAXM],
                                         double rdT, double rxh, double rsd, double rz,
                                         double ryt, int lane)
{
    const int N = P.N, M = P.M, E = P.E;
    double u = rxh;
    if (E > 0) {
        const double t = matvec<MAXM>(S.At, P.ldz, E, N, rxh, lane) - ryt;
        u = rxh - matvecT<MAXM>(S.At, P.ldz, E, N, t, lane);
    }
    const double g = rz - rsd - matvec<MAXM>(S.Gh, P.ldz, M, N, u, lane);
    return ldl_reg_solve<MAXM>(t, M, g, rdT, lane);
}

// x/y-part of solve_kkt for (the sum of) wz: dxh = -q + At^T(At q - ryt), dyt = -(At q - ryt)
template <int MAXM>
__device__ __forceinline__ void kkt_xy(const KParams &P, const Lds &S, double rxh, double ryt,
                                       double wz, int lane, double &dxh, double &dyt)
{
    const int N = P.N, M = P.M, E = P.E;
    const double q = rxh + matvecT<MAXM>(S.Gh, P.ldz, M, N, wz, lane);
    dxh = -q;
    dyt = 0.0;
    if (E > 0) {
        const double e = matvec<MAXM>(S.At, P.ldz, E, N, q, lane) - ryt;
        dxh += matvecT<MAXM>(S.At, P.ldz, E, N, e, lane);
        dyt = -e;
    }
}

// ry = dyn_res(z) with the registered device model (qp_wrapper.py:326-345): z element i on lane i.
// scratch: N + E doubles of LDS.  Rows: (T-1) n dynamics rows knot-major, then n rows x_0 - x0.
template <class Map>
__device__ __forceinline__ void knot_step(const double *zs, int t, int nt, double dt, double *fs)
{
    double xs[Map::NX], us[Map::NU], out[Map::NX];
#pragma unroll
    for (int k = 0; k < Map::NX; ++k) xs[k] = zs[t * nt + k];
#pragma unroll
    for (int k = 0; k < Map::NU; ++k) us[k] = zs[t * nt + Map::NX + k];
    Map::template step<double>(xs, us, dt, out);
#pragma unroll
    for (int k = 0; k < Map::NX; ++k) fs[t * Map::NX + k] = out[k];
}

__device__ __forceinline__ double true_dyn_res(const KParams &P, double *scratch, double z, int qp, int lane)
{
    const int n = P.dynN, m = P.dynM, T = P.dynT, nt = n + m, N = P.N, E = P.E;
    double *zs = scratch, *fs = scratch + N;
    WSYNC();
    if (lane < N) zs[lane] = z;
    WSYNC();
    if (lane < T - 1) {
        switch (P.dynId) {
        case DQP_DYN_PENDULUM1L: knot_step<dyn::Robot<dyn::Pendulum1l>>(zs, lane, nt, P.dynDt, fs); break;
        case DQP_DYN_CARTPOLE1L: knot_step<dyn::Robot<dyn::Cartpole1l>>(zs, lane, nt, P.dynDt, fs); break;
        case DQP_DYN_CARTPOLE2L: knot_step<dyn::Robot<dyn::Cartpole2l>>(zs, lane, nt, P.dynDt, fs); break;
        case DQP_DYN_PENDULUM_EULER: knot_step<dyn::PendulumEuler>(zs, lane, nt, P.dynDt, fs); break;
        case DQP_DYN_PENDULUM_DX: knot_step<dyn::PendulumDx>(zs, lane, nt, P.dynDt, fs); break;
        default: __builtin_trap();      // fill_params only lets the models above through
        }
    }
    WSYNC();
    double ry = 0.0;
    if (lane < E) {
        const int t = lane / n, i = lane - t * n;
        if (t < T - 1) ry = fs[lane] - zs[(t + 1) * nt + i];
        else if (t == T - 1) ry = zs[i] - P.dynX0[(long long)qp * n + i];
        else ry = zs[(T - 1) * nt + i];         // add_goal_constraint rows: x_{T-1} - 0 (qp_wrapper.py:339-341,650-652)
    }
    return ry;
}

__device__ __forceinline__ double step_ratio(double v, double dv, bool active)
{
    // batch.py:211-214: entries with dv > 0 never bind after the min(., 1)
    return (active && dv < 0.0) ? -v / dv : INFINITY;
}

template <int MAXM>
__global__ __launch_bounds__(WAVE) void qp_forward_kernel(KParams P)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x, qp = blockIdx.x;
    const int N = P.N, M = P.M, E = P.E;
    const Lds S = carve(sm, P);

    double rdq, rd1;
    term_zero_acc(P);
    int maxIter = P.maxIter;
    const bool batch = (P.flags & DQP_FLAG_BATCH_TERMINATION) != 0;
    const bool strict = (P.flags & DQP_FLAG_STRICT_GET_STEP) != 0;
    if (P.cap) {        // pass 2 of the batch rule: only the listed QPs, up to the reference's stop
        if (!term_flagged(P, qp)) return;
        maxIter = min(maxIter, P.cap[0]);
    }
    const int status = qp_setup(P, S, qp, lane, rdq, rd1);
    const bool inN = lane < N, inM = lane < M, inE = lane < E;

    double ph = inN ? P.p[(long long)qp * P.sp + lane] : 0.0;
    const double hh = inM ? P.h[(long long)qp * P.sh + lane] : 0.0;
    double bt = inE ? P.b[(long long)qp * P.sb + lane] : 0.0;
    ph = trsv_L(S.Lq, P.ldz, N, ph, rdq, lane);
    if (E > 0) bt = trsv_L(S.L1, P.lde, E, bt, rd1, lane);

    // initial point: d = 1, solve_kkt(p, 0, -h, -b)                     batch.py:60-74
    double t[MAXM];
    double rdT = factor_T<MAXM>(P, S, t, inM ? 1.0 : 0.0, lane);
    double xh, s, z, yt;
    {
        const double wz = kkt_wz<MAXM>(P, S, t, rdT, ph, 0.0, -hh, -bt, lane);
        kkt_xy<MAXM>(P, S, ph, -bt, wz, lane, xh, yt);
        z = wz;
        s = -wz;
    }
    {   // make s, z >= 1                                                  batch.py:76-86
        const double ms = wave_min(inM ? s : INFINITY), mz = wave_min(inM ? z : INFINITY);
        if (ms < 0.0 && inM) s -= ms - 1.0;
        if (mz < 0.0 && inM) z -= mz - 1.0;
    }

    double bxh = xh, bs = s, bz = z, byt = yt, best = INFINITY;
    bool have_best = false;
    int nNotImproved = 0, iters = 0;

    for (int it = 0; it < maxIter; ++it) {
        // residuals in hat coordinates                                   batch.py:93-108
        double rxh = xh + ph + matvecT<MAXM>(S.Gh, P.ldz, M, N, z, lane);
        double ryt = 0.0, ry = 0.0;
        if (E > 0) {
            rxh += matvecT<MAXM>(S.At, P.ldz, E, N, yt, lane);
            if (P.dynId) {      // ry = dyn_res(x) with the true model; the Newton rhs is L1^-1 ry
                const double x = trsv_LT(S.Lq, P.ldz, N, xh, rdq, lane);
                ry = true_dyn_res(P, S.T + (size_t)M * P.ldt, x, qp, lane);
                ryt = trsv_L(S.L1, P.lde, E, ry, rd1, lane);
                ryt = inE ? ryt : 0.0;
            } else {
                ryt = matvec<MAXM>(S.At, P.ldz, E, N, xh, lane) - bt;
                ry = trimatvec<MAXM>(S.L1, P.lde, E, ryt, lane);
            }
        }
        const double gx = matvec<MAXM>(S.Gh, P.ldz, M, N, xh, lane);
        const double rz = inM ? gx + s - hh : 0.0;
        const double rx = trimatvec<MAXM>(S.Lq, P.ldz, N, rxh, lane);       // rx = Lq rxh
        const double sz = wave_sum(inM ? s * z : 0.0);
        const double mu = fabs(sz / M);
        const double resid = sqrt(wave_sum(rz * rz)) + sqrt(wave_sum(ry * ry)) +
                             sqrt(wave_sum(rx * rx)) + M * mu;
        iters = it + 1;
        // best-iterate tracking                                          batch.py:119-140
        if (!have_best || resid < best) {
            nNotImproved = 0;
            have_best = true;
            best = resid; bxh = xh; bs = s; bz = z; byt = yt;
        } else {
            nNotImproved += 1;
        }
        if (batch) {                           // the stop is decided over the batch (dqp_term.hip)
            if (P.hist && lane == 0) hist_put(P, qp, it, resid, mu);
            if (!(fabs(resid) < INFINITY)) break;
        } else if ((nNotImproved >= P.notImprovedLim && best < P.stallTol) || best < P.eps ||
                   mu > 1e32 || !(fabs(resid) < INFINITY))
            break;

        const double dinv = inM ? s / z : 0.0;                          // 1/d, d = z/s
        rdT = factor_T<MAXM>(P, S, t, dinv, lane);

        // affine direction (rs = z  =>  rs/d = s)                        batch.py:151
        const double dz_a = kkt_wz<MAXM>(P, S, t, rdT, rxh, s, rz, ryt, lane);
        const double ds_a = inM ? (-z - dz_a) * dinv : 0.0;
        double alpha = fmin(wave_min(fmin(step_ratio(z, dz_a, inM), step_ratio(s, ds_a, inM))), 1.0);
        const double t3 = wave_sum(inM ? (s + alpha * ds_a) * (z + alpha * dz_a) : 0.0);
        double sig = t3 / sz;
        sig = sig * sig * sig;
        // corrector: rx = rz = ry = 0, rs = (-mu sig + ds_a dz_a)/s       batch.py:171-181
        const double rs_c = inM ? (-mu * sig + ds_a * dz_a) / s : 0.0;
        const double dz_c = ldl_reg_solve<MAXM>(t, M, -rs_c * dinv, rdT, lane);
        const double ds_c = inM ? (-rs_c - dz_c) * dinv : 0.0;
        const double dz = dz_a + dz_c, ds = ds_a + ds_c;
        double dxh, dyt;
        kkt_xy<MAXM>(P, S, rxh, ryt, dz, lane, dxh, dyt);

        // DQP_FLAG_STRICT_GET_STEP: batch.py:211-214 divides by the step; an exactly-zero component makes the
        // reference's iterate NaN, i.e. the problem keeps the best iterate it has (its history is NaN from here)
        if (strict && __builtin_amdgcn_ballot_w64(inM && (dz_a == 0.0 || ds_a == 0.0 || dz == 0.0 || ds == 0.0)) != 0)
            break;
        alpha = fmin(0.999 * wave_min(fmin(step_ratio(z, dz, inM), step_ratio(s, ds, inM))), 1.0);
        xh += alpha * dxh;
        s += alpha * ds;
        z += alpha * dz;
        yt += alpha * dyt;
    }

    if (P.hist && lane == 0) hist_fill(P, qp, iters);
    // back to the caller's coordinates: x = Lq^-T xh, y = L1^-T yt
    const double x = trsv_LT(S.Lq, P.ldz, N, bxh, rdq, lane);
    if (inN) P.zhat[(long long)qp * N + lane] = x;
    if (inM) {
        P.lam[(long long)qp * M + lane] = bz;
        P.slack[(long long)qp * M + lane] = bs;
    }
    if (E > 0) {
        const double y = trsv_LT(S.L1, P.lde, E, byt, rd1, lane);
        if (inE) P.nu[(long long)qp * E + lane] = y;
    }
    if (lane == 0) {
        if (P.info) { P.info[2 * qp] = status; P.info[2 * qp + 1] = iters; }
        if (P.best_resid) P.best_resid[qp] = best;
    }
}

// QPFunctionFn.backward (qp.py:128-183) / DenseQPFunction Solver.backward (qp.py:239-270)
template <int MAXM>
__global__ __launch_bounds__(WAVE) void qp_backward_kernel(KParams P)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x, qp = blockIdx.x;
    const int N = P.N, M = P.M, E = P.E;
    const Lds S = carve(sm, P);

    double rdq, rd1;
    const int status = qp_setup(P, S, qp, lane, rdq, rd1);
    const bool inN = lane < N, inM = lane < M, inE = lane < E;

    const double zh = inN ? P.zin[(long long)qp * N + lane] : 0.0;
    const double lam = inM ? P.lamin[(long long)qp * M + lane] : 0.0;
    const double slk = inM ? P.slackin[(long long)qp * M + lane] : 1.0;
    const double nu = inE ? P.nuin[(long long)qp * E + lane] : 0.0;
    double g = inN ? P.gin[(long long)qp * N + lane] : 0.0;

    double dinv;
    if (P.flags & DQP_FLAG_DENSE_BACKWARD) dinv = inM ? slk / lam : 0.0;
    else dinv = inM ? fmax(slk, 1e-8) / fmax(lam, 1e-8) : 0.0;       // qp.py:149
    double t[MAXM];
    const double rdT = factor_T<MAXM>(P, S, t, dinv, lane);

    // solve_kkt(rx = dl_dzhat, 0, 0, 0)
    const double rxh = trsv_L(S.Lq, P.ldz, N, g, rdq, lane);
    const double dlam = kkt_wz<MAXM>(P, S, t, rdT, rxh, 0.0, 0.0, 0.0, lane);
    double dxh, dyt;
    kkt_xy<MAXM>(P, S, rxh, 0.0, dlam, lane, dxh, dyt);
    const double dx = trsv_LT(S.Lq, P.ldz, N, dxh, rdq, lane);
    double dnu = 0.0;
    if (E > 0) dnu = trsv_LT(S.L1, P.lde, E, dyt, rd1, lane);

    // gradients; lanes run along the nz (contiguous) axis -> coalesced row stores
    if (P.dp && inN) P.dp[(long long)qp * N + lane] = dx;
    if (P.dh && inM) P.dh[(long long)qp * M + lane] = -dlam;
    if (P.db && inE) P.db[(long long)qp * E + lane] = -dnu;
    if (P.dQ) {
        double *o = P.dQ + (long long)qp * N * N;
        for (int i = 0; i < N; ++i) {
            const double dxi = bcast(dx, i), zi = bcast(zh, i);
            if (inN) o[i * N + lane] = 0.5 * (dxi * zh + zi * dx);
        }
    }
    if (P.dG) {
        double *o = P.dG + (long long)qp * M * N;
        for (int i = 0; i < M; ++i) {
            const double dli = bcast(dlam, i), li = bcast(lam, i);
            if (inN) o[i * N + lane] = dli * zh + li * dx;
        }
    }
    if (P.dA && E > 0) {
        double *o = P.dA + (long long)qp * E * N;
        for (int i = 0; i < E; ++i) {
            const double dni = bcast(dnu, i), ni = bcast(nu, i);
            if (inN) o[i * N + lane] = dni * zh + ni * dx;
        }
    }
    if (lane == 0 && P.info) { P.info[2 * qp] = status; P.info[2 * qp + 1] = 0; }
}

static inline bool is_big(const KParams &P) { return P.N > DQP_MAX_DIM || P.M > DQP_MAX_DIM || P.E > DQP_MAX_DIM; }

void fill_opts(const dqp_opts *o, KParams &P)
{
    P.eps = (o ? o->eps : 1e-12) * 0.1;
    P.stallTol = o ? o->stall_tol : 1e-10;
    P.maxIter = o ? o->max_iter : 20;
    P.notImprovedLim = o ? o->not_improved_lim : 3;
    P.flags = o ? o->flags : 0u;
    P.hist = nullptr; P.cap = nullptr; P.histIters = P.maxIter;
}

int fill_params(const dqp_dims *d, const dqp_opts *o, KParams &P, size_t &lds_bytes)
{
#ifdef DQP_STAMPS
    P.stamps = dqp::g_debug_stamps;
#else
    P.stamps = nullptr;
#endif
    if (!d) return DQP_ERR_BAD_ARG;
    if (d->nbatch < 0 || d->nz <= 0 || d->nineq <= 0 || d->neq < 0) return DQP_ERR_BAD_ARG;
    if (d->nz > DQP_MAX_DIM_LARGE || d->nineq > DQP_MAX_DIM_LARGE || d->neq > DQP_MAX_DIM_LARGE) return DQP_ERR_TOO_LARGE;
    P.B = d->nbatch; P.N = d->nz; P.M = d->nineq; P.E = d->neq;
    if (is_big(P)) {        // one QP per workgroup, matrices in the caller's workspace (dqp_big.hip)
        if (!big_fits(P.N, P.M, P.E)) return DQP_ERR_TOO_LARGE;
        if (o && o->dyn_id) return DQP_ERR_BAD_ARG;
        P.sQ = d->stride_Q; P.sp = d->stride_p; P.sG = d->stride_G;
        P.sh = d->stride_h; P.sA = d->stride_A; P.sb = d->stride_b;
        fill_opts(o, P);
        lds_bytes = 0;
        return DQP_OK;
    }
    P.ldz = d->nz | 1; P.ldm = d->nineq | 1; P.lde = d->neq | 1;
    P.ldt = P.ldz > P.ldm ? P.ldz : P.ldm;
    P.sQ = d->stride_Q; P.sp = d->stride_p; P.sG = d->stride_G;
    P.sh = d->stride_h; P.sA = d->stride_A; P.sb = d->stride_b;
    // per-problem exit threshold: one decade below the reference's batch-wide eps (include/dqp.h):
    // a large batch never meets `best_resids.max() < eps` before maxIter, so the reference keeps
    // polishing every problem; stopping exactly at eps left mu ~10x larger than the reference's
    // and moved the gradients of weakly active constraints (lam ~ 1e-4) by up to 1e-3 relative
    // (tools/stress_parity.py: 58 tolerance exceedances in 147k QPs -> 10 with the extra decade,
    // the same as never stopping at eps at all; costs ~1 iteration on average).
    fill_opts(o, P);
    P.dynId = o ? o->dyn_id : 0;
    P.dynT = 0; P.dynN = 0; P.dynM = 0; P.dynDt = 0.0; P.dynX0 = nullptr;
    if (P.dynId) {
        int32_t n = 0, m = 0;
        if (dqp_dyn_sizes(P.dynId, &n, &m) != DQP_OK) return DQP_ERR_BAD_ARG;
        // models true_dyn_res evaluates on these one-QP-per-wavefront kernels (the quadrotor runs on the stage-wise ones)
        if (P.dynId != DQP_DYN_PENDULUM1L && P.dynId != DQP_DYN_CARTPOLE1L && P.dynId != DQP_DYN_CARTPOLE2L &&
            P.dynId != DQP_DYN_PENDULUM_EULER && P.dynId != DQP_DYN_PENDULUM_DX)
            return DQP_ERR_BAD_ARG;
        P.dynT = o->dyn_T; P.dynN = n; P.dynM = m; P.dynDt = o->dyn_dt; P.dynX0 = o->dyn_x0;
        if (P.dynT < 2 || P.dynT > WAVE || P.N != P.dynT * (n + m) ||
            (P.E != P.dynT * n && P.E != (P.dynT + 1) * n) || !P.dynX0)
            return DQP_ERR_BAD_ARG;
    }
    size_t n = (size_t)P.N * P.ldz + (size_t)P.M * P.ldz + (size_t)P.E * P.ldz +
               (size_t)P.E * P.lde + (size_t)P.M * (P.ldm > P.lde ? P.ldm : P.lde) +
               (size_t)P.M * P.ldt;
    if (P.dynId) n += (size_t)P.N + P.E;        // true_dyn_res scratch behind the T buffer
    lds_bytes = n * sizeof(double);
    if (lds_bytes > 160 * 1024) return DQP_ERR_TOO_LARGE;
    return DQP_OK;
}

template <typename K>
int launch(K kernel, const KParams &P, size_t lds_bytes, void *stream)
{
    if (P.B == 0) return DQP_OK;
    if (lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return DQP_ERR_LAUNCH;
    }
    DQP_LAUNCH(kernel, dim3(P.B), dim3(WAVE), lds_bytes, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // namespace

#ifdef DQP_STAMPS
namespace dqp { unsigned long long *g_debug_stamps = nullptr; }
#endif

extern "C" {

#ifdef DQP_STAMPS
// Diagnostic hook of the instrumented build only (tools/stamps.py compiles libdqp_hip_stamps.so
// with -DDQP_STAMPS; the shipped library has neither this symbol nor any state): device buffer
// of 16 x uint64 per workgroup that the DPP-row kernels fill with s_memtime stamps.
__attribute__((visibility("default"))) void dqp_debug_set_stamps(void *dev_ptr)
{
    dqp::g_debug_stamps = (unsigned long long *)dev_ptr;
}
#endif

__attribute__((visibility("default"))) int dqp_version(void) { return DQP_VERSION; }

__attribute__((visibility("default"))) const char *dqp_error_string(int code)
{
    switch (code) {
    case DQP_OK: return "ok";
    case DQP_ERR_BAD_ARG: return "bad argument";
    case DQP_ERR_TOO_LARGE: return "problem too large for the fused one-wavefront kernels";
    case DQP_ERR_LAUNCH: return "HIP kernel launch failed";
    case DQP_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown error";
    }
}

__attribute__((visibility("default"))) size_t dqp_workspace_bytes(const dqp_dims *d)
{
    if (!d || d->nbatch <= 0) return 0;
    if (d->nz > DQP_MAX_DIM || d->nineq > DQP_MAX_DIM || d->neq > DQP_MAX_DIM) {        // required at these sizes
        if (d->nz > DQP_MAX_DIM_LARGE || d->nineq > DQP_MAX_DIM_LARGE || d->neq > DQP_MAX_DIM_LARGE) return 0;
        return (size_t)d->nbatch * (size_t)dqp::big_workspace_doubles(d->nz, d->nineq, d->neq) * sizeof(double);
    }
    return (size_t)d->nbatch * (size_t)dqp::r16n_workspace_doubles(d->nz, d->nineq, d->neq) * sizeof(double);
}

__attribute__((visibility("default"))) size_t
dqp_termination_bytes(const dqp_dims *d, const dqp_opts *o)
{
    if (!d || d->nbatch <= 0 || !o || !(o->flags & DQP_FLAG_BATCH_TERMINATION)) return 0;
    return dqp::term_bytes(d->nbatch, o->max_iter, dqp::r16n_snapshot_doubles(d->nz, d->nineq, d->neq));
}

static int forward_once(const KParams &P, size_t lds, void *workspace, void *stream);

__attribute__((visibility("default"))) int
dqp_qp_forward(const dqp_dims *dims, const dqp_opts *opts, const double *Q, const double *p,
               const double *G, const double *h, const double *A, const double *b, double *zhat,
               double *lam, double *nu, double *slack, int32_t *info, double *best_resid,
               void *workspace, void *termination, void *stream)
{
    KParams P = {};
    size_t lds = 0;
    int rc = fill_params(dims, opts, P, lds);
    if (rc != DQP_OK) return rc;
    if (P.B == 0) return DQP_OK;
    if (!Q || !p || !G || !h || !zhat || !lam || !slack) return DQP_ERR_BAD_ARG;
    P.workspace = (double *)workspace;
    if (P.E > 0 && (!A || !b || !nu)) return DQP_ERR_BAD_ARG;
    P.Q = Q; P.p = p; P.G = G; P.h = h; P.A = A; P.b = b;
    P.zhat = zhat; P.lam = lam; P.nu = nu; P.slack = slack;
    P.info = info; P.best_resid = best_resid;
    if (!(P.flags & DQP_FLAG_BATCH_TERMINATION)) return forward_once(P, lds, workspace, stream);
    // The reference's batch-coupled stop (batch.py:119-144), replayed on the device:
    //   1. every problem iterates to max_iter, recording (resid, mu) per iteration;
    //   2. the batch rule is evaluated on that history -> the iteration the reference stops at,
    //      and the list of problems whose best iterate came after it;
    //   3. those problems (none in a large batch) are solved again up to that iteration.
    // Three enqueues on `stream`, no host synchronisation, hipGraph-capturable.
    if (!termination || P.maxIter < 1 || P.maxIter > 64) return DQP_ERR_BAD_ARG;
    P.eps = opts ? opts->eps : 1e-12;            // the batch rule uses the reference's eps itself
    // null-space kernels keep their improving iterates: pass 2 is then an epilogue, not a re-solve
    const bool nullspace = workspace && !(P.flags & (DQP_FLAG_GENERIC_ONLY | DQP_FLAG_NO_NULLSPACE)) && !P.dynId && !is_big(P);
    term_bind_pass1(P, termination, nullspace ? r16n_snapshot_doubles(P.N, P.M, P.E) : 0);
    if ((rc = forward_once(P, lds, workspace, stream)) != DQP_OK) return rc;
    if (P.flags & DQP_FLAG_HISTORY_ONLY) return DQP_OK;
    if ((rc = term_decide(P, termination, stream)) != DQP_OK) return rc;
    term_bind_pass2(P, termination);
    return forward_once(P, lds, workspace, stream);
}

// ---- the batch rule over a batch that is sharded across devices (include/dqp.h)
__attribute__((visibility("default"))) int
dqp_term_local_masks(const dqp_dims *dims, const dqp_opts *opts, void *termination, uint64_t *masks, void *stream)
{
    KParams P = {};
    size_t lds = 0;
    int rc = fill_params(dims, opts, P, lds);
    if (rc != DQP_OK) return rc;
    if (!(P.flags & DQP_FLAG_BATCH_TERMINATION) || !termination || !masks || P.maxIter < 1 || P.maxIter > 64)
        return DQP_ERR_BAD_ARG;
    if (P.B == 0) return hipMemsetAsync(masks, 0, 24, (hipStream_t)stream) == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
    P.eps = opts ? opts->eps : 1e-12;
    return term_local_masks(P, termination, (unsigned long long *)masks, stream);
}

__attribute__((visibility("default"))) int
dqp_qp_forward_finish(const dqp_dims *dims, const dqp_opts *opts, const double *Q, const double *p,
                      const double *G, const double *h, const double *A, const double *b, double *zhat,
                      double *lam, double *nu, double *slack, int32_t *info, double *best_resid,
                      void *workspace, void *termination, const uint64_t *masks, void *stream)
{
    KParams P = {};
    size_t lds = 0;
    int rc = fill_params(dims, opts, P, lds);
    if (rc != DQP_OK) return rc;
    if (P.B == 0) return DQP_OK;
    if (!(P.flags & DQP_FLAG_BATCH_TERMINATION) || !termination || !masks || P.maxIter < 1 || P.maxIter > 64)
        return DQP_ERR_BAD_ARG;
    if (!Q || !p || !G || !h || !zhat || !lam || !slack) return DQP_ERR_BAD_ARG;
    if (P.E > 0 && (!A || !b || !nu)) return DQP_ERR_BAD_ARG;
    P.workspace = (double *)workspace;
    P.Q = Q; P.p = p; P.G = G; P.h = h; P.A = A; P.b = b;
    P.zhat = zhat; P.lam = lam; P.nu = nu; P.slack = slack;
    P.info = info; P.best_resid = best_resid;
    P.eps = opts ? opts->eps : 1e-12;
    const bool nullspace = workspace && !(P.flags & (DQP_FLAG_GENERIC_ONLY | DQP_FLAG_NO_NULLSPACE)) && !P.dynId && !is_big(P);
    term_bind_pass1(P, termination, nullspace ? r16n_snapshot_doubles(P.N, P.M, P.E) : 0);    // (snapshot pointer)
    if ((rc = term_decide_global(P, termination, (const unsigned long long *)masks, stream)) != DQP_OK) return rc;
    term_bind_pass2(P, termination);
    return forward_once(P, lds, workspace, stream);
}

static int forward_once(const KParams &P, size_t lds, void *workspace, void *stream)
{
    int rc;
    if (is_big(P)) return workspace ? big_forward(P, stream) : DQP_ERR_BAD_ARG;
    if (!(P.flags & DQP_FLAG_GENERIC_ONLY) && !P.dynId) {
        // DPP-row kernels for the instantiated sizes: the null-space form when the caller gave
        // it its workspace (and did not opt out), else the form that keeps the equality rows.
        rc = (workspace && !(P.flags & DQP_FLAG_NO_NULLSPACE)) ? r16n_forward(P, stream) : 1;
        if (rc == 1) rc = r16_forward(P, stream);
        if (rc != 1) return rc;
    }
    const int mx = P.N > P.M ? (P.N > P.E ? P.N : P.E) : (P.M > P.E ? P.M : P.E);
    if (mx <= 16) return launch(qp_forward_kernel<16>, P, lds, stream);
    if (mx <= 32) return launch(qp_forward_kernel<32>, P, lds, stream);
    return launch(qp_forward_kernel<64>, P, lds, stream);
}

// MPC-structured entry points: the null-space kernels assemble (Q,p,G,h,A,b) in registers from the
// time-major MPC data and scatter the gradients back the same way -- no dense QP in HBM.
// Which kernels serve an MPC shape: the dense null-space kernels where the QP size has an
// instantiation (small horizons: one QP in registers), else the stage-wise Riccati kernels
// (dqp_ric.hip: any horizon, n + m <= 16).
enum { MPC_NONE = 0, MPC_R16N = 1, MPC_RIC = 2 };

static int mpc_params(const dqp_mpc_dims *md, const dqp_opts *opts, KParams &P, size_t &lds, int &kind)
{
    kind = MPC_NONE;
    if (!md || md->T < 2 || md->n_state < 1 || md->n_ctrl < 1 || !md->has_bounds || md->nbatch < 0) return DQP_ERR_BAD_ARG;
    dqp_dims d = {};
    d.nbatch = md->nbatch;
    d.nz = md->T * (md->n_state + md->n_ctrl);
    d.nineq = 2 * md->T * md->n_ctrl;
    d.neq = md->T * md->n_state;
    if (md->dyn_id) {       // true-dynamics residual: a registered model of this size, stage-wise kernels
        int32_t dn = 0, dm = 0;
        if (dqp_dyn_sizes(md->dyn_id, &dn, &dm) != DQP_OK || dn != md->n_state || dm != md->n_ctrl) return DQP_ERR_BAD_ARG;
    }
    const bool stagewise = opts && (opts->flags & DQP_FLAG_STAGEWISE);
    if (!md->dyn_id && !stagewise && r16n_workspace_doubles(d.nz, d.nineq, d.neq) > 0) {
        dqp_opts o2;
        if (opts) { o2 = *opts; o2.dyn_id = 0; }
        int rc = fill_params(&d, opts ? &o2 : nullptr, P, lds);
        if (rc != DQP_OK) return rc;
        kind = MPC_R16N;
    } else if (ric_supported(md->n_state, md->n_ctrl)) {
        // the stage-wise kernels address a wavefront's four workspaces with 32-bit byte offsets
        if (md->T > 200000 || ric_workspace_doubles(md->n_state, md->n_ctrl, md->T) * 8 * 4 > 0x7fffffffLL) return DQP_ERR_TOO_LARGE;
        P.stamps = nullptr;
        P.B = d.nbatch; P.N = d.nz; P.M = d.nineq; P.E = d.neq;
        fill_opts(opts, P);
        P.dynId = md->dyn_id;
        P.dynDt = opts ? opts->dyn_dt : 0.0;
        kind = MPC_RIC;
    } else {
        return DQP_ERR_TOO_LARGE;
    }
    P.mn = md->n_state; P.mm = md->n_ctrl; P.mT = md->T;
    return DQP_OK;
}

static size_t mpc_workspace_doubles(const KParams &P, int kind)
{
    if (kind == MPC_R16N) return (size_t)P.B * (size_t)r16n_workspace_doubles(P.N, P.M, P.E);
    // four problems per wavefront, padding rows own a slot too
    return (size_t)((P.B + 3) / 4 * 4) * (size_t)ric_workspace_doubles(P.mn, P.mm, P.mT);
}

static int mpc_snapshot_doubles(const KParams &P, int kind)
{
    return kind == MPC_R16N ? r16n_snapshot_doubles(P.N, P.M, P.E) : ric_snapshot_doubles(P.mn, P.mm, P.mT);
}

__attribute__((visibility("default"))) size_t dqp_mpc_qp_termination_bytes(const dqp_mpc_dims *md, const dqp_opts *o)
{
    KParams P = {};
    size_t lds = 0;
    int kind;
    if (!o || !(o->flags & DQP_FLAG_BATCH_TERMINATION) || mpc_params(md, o, P, lds, kind) != DQP_OK || md->nbatch <= 0) return 0;
    return term_bytes(P.B, P.maxIter, mpc_snapshot_doubles(P, kind));
}

__attribute__((visibility("default"))) int dqp_mpc_qp_supported(const dqp_mpc_dims *md)
{
    KParams P = {};
    size_t lds = 0;
    int kind;
    return mpc_params(md, nullptr, P, lds, kind) == DQP_OK ? 1 : 0;
}

__attribute__((visibility("default"))) size_t dqp_mpc_qp_workspace_bytes(const dqp_mpc_dims *md)
{
    KParams P = {};
    size_t lds = 0;
    int kind;
    if (mpc_params(md, nullptr, P, lds, kind) != DQP_OK || md->nbatch <= 0) return 0;
    return mpc_workspace_doubles(P, kind) * sizeof(double);
}

__attribute__((visibility("default"))) int
dqp_mpc_qp_forward(const dqp_mpc_dims *md, const dqp_opts *opts, const double *C, const double *c,
                   const double *F, const double *f, const double *x0, const double *u_lower,
                   const double *u_upper, double *tau, double *lam, double *nu, double *slack,
                   int32_t *info, double *best_resid, void *workspace, void *termination, void *stream)
{
    KParams P = {};
    size_t lds = 0;
    int kind;
    int rc = mpc_params(md, opts, P, lds, kind);
    if (rc != DQP_OK) return rc;
    if (P.B == 0) return DQP_OK;
    if (!C || !c || !F || !f || !x0 || !u_lower || !u_upper || !tau || !lam || !nu || !slack || !workspace)
        return DQP_ERR_BAD_ARG;
    if (P.dynId && !(P.dynDt > 0.0)) return DQP_ERR_BAD_ARG;        // the model's step needs dqp_opts.dyn_dt
    auto run = [&](const KParams &Q) { return kind == MPC_R16N ? r16n_forward(Q, stream) : ric_forward(Q, stream); };
    P.mC = C; P.mc = c; P.mF = F; P.mf = f; P.mx0 = x0; P.mul = u_lower; P.muu = u_upper;
    P.zhat = tau; P.lam = lam; P.nu = nu; P.slack = slack; P.info = info; P.best_resid = best_resid;
    P.workspace = (double *)workspace;
    if (P.maxIter < 1) return DQP_ERR_BAD_ARG;
    if (!(P.flags & DQP_FLAG_BATCH_TERMINATION)) return run(P);
    if (!termination || P.maxIter > 64) return DQP_ERR_BAD_ARG;
    P.eps = opts ? opts->eps : 1e-12;
    term_bind_pass1(P, termination, mpc_snapshot_doubles(P, kind));
    if ((rc = run(P)) != DQP_OK) return rc;
    if (P.flags & DQP_FLAG_HISTORY_ONLY) return DQP_OK;
    if ((rc = term_decide(P, termination, stream)) != DQP_OK) return rc;
    term_bind_pass2(P, termination);
    return kind == MPC_R16N ? r16n_forward(P, stream) : ric_finish(P, stream);
}

// the stage-wise kernels only, whatever the horizon (the null-space kernels keep their iterate in registers)
static int stepped_params(const dqp_mpc_dims *md, const dqp_opts *opts, KParams &P)
{
    if (!md || md->nbatch < 0 || md->n_state <= 0 || md->n_ctrl <= 0 || md->T < 2 || !md->has_bounds) return DQP_ERR_BAD_ARG;
    if (!ric_supported(md->n_state, md->n_ctrl)) return DQP_ERR_TOO_LARGE;
    if (md->T > 200000 || ric_workspace_doubles(md->n_state, md->n_ctrl, md->T) * 8 * 4 > 0x7fffffffLL) return DQP_ERR_TOO_LARGE;
    P.B = md->nbatch;
    P.N = md->T * (md->n_state + md->n_ctrl); P.M = 2 * md->T * md->n_ctrl; P.E = md->T * md->n_state;
    fill_opts(opts, P);
    P.mn = md->n_state; P.mm = md->n_ctrl; P.mT = md->T;
    return DQP_OK;
}

__attribute__((visibility("default"))) size_t dqp_mpc_qp_stepped_workspace_bytes(const dqp_mpc_dims *md)
{
    KParams P = {};
    if (stepped_params(md, nullptr, P) != DQP_OK || md->nbatch <= 0) return 0;
    return (size_t)ric_stepped_workspace_doubles(P.mn, P.mm, P.mT, P.B) * sizeof(double);
}

__attribute__((visibility("default"))) size_t dqp_mpc_qp_stepped_termination_bytes(const dqp_mpc_dims *md, const dqp_opts *o)
{
    KParams P = {};
    if (!o || !(o->flags & DQP_FLAG_BATCH_TERMINATION) || stepped_params(md, o, P) != DQP_OK || md->nbatch <= 0) return 0;
    return term_bytes(P.B, P.maxIter, ric_snapshot_doubles(P.mn, P.mm, P.mT));
}

__attribute__((visibility("default"))) int
dqp_mpc_qp_forward_stepped(const dqp_mpc_dims *md, const dqp_opts *opts, const double *C, const double *c,
                           const double *F, const double *f, const double *x0, const double *u_lower,
                           const double *u_upper, const double *ext_ry, int32_t it_begin, int32_t it_end,
                           double *tau, double *lam, double *nu, double *slack, int32_t *info, double *best_resid,
                           void *workspace, void *termination, void *stream)
{
    KParams P = {};
    int rc = stepped_params(md, opts, P);
    if (rc != DQP_OK) return rc;
    if (P.B == 0) return DQP_OK;
    if (!C || !c || !F || !f || !x0 || !u_lower || !u_upper || !tau || !lam || !nu || !slack || !workspace)
        return DQP_ERR_BAD_ARG;
    if (P.maxIter < 1 || it_begin < 0 || it_end < it_begin || it_end > P.maxIter || it_end - it_begin > 1) return DQP_ERR_BAD_ARG;
    if (it_end > it_begin && !ext_ry) return DQP_ERR_BAD_ARG;
    if (it_end == it_begin && it_begin != 0) return DQP_ERR_BAD_ARG;
    P.mC = C; P.mc = c; P.mF = F; P.mf = f; P.mx0 = x0; P.mul = u_lower; P.muu = u_upper;
    P.zhat = tau; P.lam = lam; P.nu = nu; P.slack = slack; P.info = info; P.best_resid = best_resid;
    P.workspace = (double *)workspace;
    P.extRy = ext_ry; P.itBegin = it_begin; P.itEnd = it_end;
    const bool batch = (P.flags & DQP_FLAG_BATCH_TERMINATION) != 0;
    if (batch) {
        if (!termination || P.maxIter > 64) return DQP_ERR_BAD_ARG;
        P.eps = opts ? opts->eps : 1e-12;
        term_bind_pass1(P, termination, ric_snapshot_doubles(P.mn, P.mm, P.mT));
    }
    if ((rc = ric_forward_stepped(P, stream)) != DQP_OK) return rc == 1 ? DQP_ERR_TOO_LARGE : rc;
    if (it_end < P.maxIter || !batch || (P.flags & DQP_FLAG_HISTORY_ONLY)) return DQP_OK;
    if ((rc = term_decide(P, termination, stream)) != DQP_OK) return rc;
    term_bind_pass2(P, termination);
    return ric_finish(P, stream);
}

__attribute__((visibility("default"))) int
dqp_mpc_qp_backward(const dqp_mpc_dims *md, const dqp_opts *opts, const double *C, const double *F,
                    const double *tau, const double *lam, const double *nu, const double *slack,
                    const double *dl_dtau, double *dC, double *dc, double *dF, double *df, double *dx0,
                    int32_t *info, void *workspace, void *stream)
{
    KParams P = {};
    size_t lds = 0;
    int kind;
    int rc = mpc_params(md, opts, P, lds, kind);
    if (rc != DQP_OK) return rc;
    if (P.B == 0) return DQP_OK;
    if (!tau || !lam || !nu || !slack || !dl_dtau || !workspace) return DQP_ERR_BAD_ARG;
    if (!dC && !dc && !dF && !df && !dx0) return DQP_OK;
    P.zin = tau; P.lamin = lam; P.nuin = nu; P.slackin = slack; P.gin = dl_dtau;
    P.mdC = dC; P.mdc = dc; P.mdF = dF; P.mdf = df; P.mdx0 = dx0;
    P.info = info;
    P.workspace = (double *)workspace;
    if (kind == MPC_R16N) return r16n_backward(P, stream);
    if (!C || !F) return DQP_ERR_BAD_ARG;
    P.mC = C; P.mF = F;
    return ric_backward(P, stream);
}

__attribute__((visibility("default"))) int
dqp_qp_backward(const dqp_dims *dims, const dqp_opts *opts, const double *Q, const double *G,
                const double *A, const double *zhat, const double *lam, const double *nu,
                const double *slack, const double *dl_dzhat, double *dQ, double *dp, double *dG,
                double *dh, double *dA, double *db, int32_t *info, void *workspace,
                void *stream)
{
    KParams P = {};
    size_t lds = 0;
    int rc = fill_params(dims, opts, P, lds);
    if (rc != DQP_OK) return rc;
    if (P.B == 0) return DQP_OK;
    if (!Q || !G || !zhat || !lam || !slack || !dl_dzhat) return DQP_ERR_BAD_ARG;
    if (P.E > 0 && (!A || !nu)) return DQP_ERR_BAD_ARG;
    P.Q = Q; P.G = G; P.A = A;
    P.zin = zhat; P.lamin = lam; P.nuin = nu; P.slackin = slack; P.gin = dl_dzhat;
    P.dQ = dQ; P.dp = dp; P.dG = dG; P.dh = dh; P.dA = dA; P.db = db;
    P.info = info;
    P.workspace = (double *)workspace;
    if (is_big(P)) return workspace ? big_backward(P, stream) : DQP_ERR_BAD_ARG;
    if (!(P.flags & DQP_FLAG_GENERIC_ONLY)) {
        // Backward is ONE solve, always in the Schur-complement form (T = Gz Gz^T + D^-1 keeps
        // dlam accurate for strongly active constraints, d ~ 1e8 after the reference's clamps).
        // With the forward's workspace (DQP_FLAG_BACKWARD_CTX) nothing is refactored.
        rc = ((P.flags & DQP_FLAG_BACKWARD_CTX) && workspace && !(P.flags & DQP_FLAG_NO_NULLSPACE))
                 ? r16n_backward(P, stream) : 1;
        if (rc == 1) rc = r16_backward(P, stream);
        if (rc != 1) return rc;
    }
    const int mx = P.N > P.M ? (P.N > P.E ? P.N : P.E) : (P.M > P.E ? P.M : P.E);
    if (mx <= 16) return launch(qp_backward_kernel<16>, P, lds, stream);
    if (mx <= 32) return launch(qp_backward_kernel<32>, P, lds, stream);
    return launch(qp_backward_kernel<64>, P, lds, stream);
}

}  // extern "C"
