// dqp_mpc.hip -- MPC-structured QP assembly on the device.
//
// Replaces qpth/qp_wrapper.py:638-679 (compute_Qq_dense / compute_Ab_dense / compute_Gh_dense),
// which build (Q,p,G,h,A,b) from the time-major MPC data with advanced-index scatters into
// freshly zeroed tensors (three fills + five scatters per call).  Here one launch writes every
// element of the six outputs exactly once (coalesced along the contiguous nz axis), and the
// backward launch gathers the gradients of (C,c,F,f,x0) from (dQ,dp,dA,db).
//
// Layouts (fp64, contiguous):  C (T,B,nt,nt)  c (T,B,nt)  F (T-1,B,n,nt)  f (T-1,B,n)  x0 (B,n)
//   z = per-timestep [x_t, u_t], nt = n+m, nz = T nt, neq = T n, nineq = 2 T m (or m without bounds)
//   A rows: (T-1) n dynamics rows  F_t [x_t;u_t] - x_{t+1} = -f_t, then n rows  x_0 = x0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dqp.h"
#include "dqp_dyn_models.h"
#include "dqp_trace.h"

namespace {

struct MpcP {
    const double *C, *c, *F, *f, *x0, *ul, *uu;
    double *Q, *p, *G, *h, *A, *b;
    const double *dQ, *dp, *dA, *db;
    double *dC, *dc, *dF, *df, *dx0;
    int B, n, m, T, bounds;
};

__global__ __launch_bounds__(256) void assemble_kernel(MpcP P)
{
    const int n = P.n, m = P.m, T = P.T, nt = n + m, nz = T * nt, neq = T * n;
    const int nineq = P.bounds ? 2 * T * m : m;
    const long long B = P.B;
    const long long nQ = B * nz * nz, nA = B * neq * nz, nG = B * nineq * nz;
    const long long np = B * nz, nb = B * neq, nh = B * nineq;
    const long long total = nQ + nA + nG + np + nb + nh;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long e = idx;
        if (e < nQ) {                                   // Q = blockdiag_t C[t,b]
            const int col = e % nz; const long long rb = e / nz;
            const int row = rb % nz; const long long b = rb / nz;
            const int t = row / nt, i = row - t * nt, tc = col / nt, j = col - tc * nt;
            P.Q[e] = (t == tc) ? P.C[(((long long)t * B + b) * nt + i) * nt + j] : 0.0;
            continue;
        }
        e -= nQ;
        if (e < nA) {
            const int col = e % nz; const long long rb = e / nz;
            const int row = rb % neq; const long long b = rb / neq;
            const int t = row / n, i = row - t * n, tc = col / nt, j = col - tc * nt;
            double v = 0.0;
            if (t < T - 1) {
                if (tc == t) v = P.F[(((long long)t * B + b) * n + i) * nt + j];
                else if (tc == t + 1 && j == i) v = -1.0;
            } else if (tc == 0 && j == i) v = 1.0;
            P.A[e] = v;
            continue;
        }
        e -= nA;
        if (e < nG) {
            const int col = e % nz; const long long rb = e / nz;
            const int row = rb % nineq;
            const int tc = col / nt, j = col - tc * nt;
            double v = 0.0;
            if (P.bounds) {
                const int k = row < T * m ? row : row - T * m;      // k = t m + ju
                const int t = k / m, ju = k - t * m;
                if (tc == t && j == n + ju) v = row < T * m ? 1.0 : -1.0;
            } else if (tc == T - 1 && j == n + row) v = 1.0;        // qp_wrapper.py:669-671
            P.G[e] = v;
            continue;
        }
        e -= nG;
        if (e < np) {                                   // p = concat_t c[t,b]
            const int col = e % nz; const long long b = e / nz;
            const int t = col / nt, j = col - t * nt;
            P.p[e] = P.c[((long long)t * B + b) * nt + j];
            continue;
        }
        e -= np;
        if (e < nb) {                                   // b = [-f ; x0]
            const int row = e % neq; const long long b = e / neq;
            const int t = row / n, i = row - t * n;
            P.b[e] = (t < T - 1) ? -P.f[((long long)t * B + b) * n + i] : P.x0[b * n + i];
            continue;
        }
        e -= nb;
        {                                               // h = [u_upper ; -u_lower] (or ones)
            const int row = e % nineq;
            double v = 1.0;
            if (P.bounds) {
                const int k = row < T * m ? row : row - T * m;
                const int ju = k % m;
                v = row < T * m ? P.uu[ju] : -P.ul[ju];
            }
            P.h[e] = v;
        }
    }
}

__global__ __launch_bounds__(256) void assemble_backward_kernel(MpcP P)
{
    const int n = P.n, m = P.m, T = P.T, nt = n + m, nz = T * nt, neq = T * n;
    const long long B = P.B;
    const long long nC = (long long)T * B * nt * nt, nc = (long long)T * B * nt;
    const long long nF = (long long)(T - 1) * B * n * nt, nf = (long long)(T - 1) * B * n, nx = B * n;
    const long long total = nC + nc + nF + nf + nx;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long e = idx;
        if (e < nC) {
            const int j = e % nt; long long q = e / nt;
            const int i = q % nt; q /= nt;
            const long long b = q % B; const int t = q / B;
            if (P.dC) P.dC[e] = P.dQ ? P.dQ[(b * nz + t * nt + i) * nz + t * nt + j] : 0.0;
            continue;
        }
        e -= nC;
        if (e < nc) {
            const int j = e % nt; long long q = e / nt;
            const long long b = q % B; const int t = q / B;
            if (P.dc) P.dc[e] = P.dp ? P.dp[b * nz + t * nt + j] : 0.0;
            continue;
        }
        e -= nc;
        if (e < nF) {
            const int j = e % nt; long long q = e / nt;
            const int i = q % n; q /= n;
            const long long b = q % B; const int t = q / B;
            if (P.dF) P.dF[e] = P.dA ? P.dA[(b * neq + t * n + i) * nz + t * nt + j] : 0.0;
            continue;
        }
        e -= nF;
        if (e < nf) {
            const int i = e % n; long long q = e / n;
            const long long b = q % B; const int t = q / B;
            if (P.df) P.df[e] = P.db ? -P.db[b * neq + t * n + i] : 0.0;
            continue;
        }
        e -= nf;
        {
            const int i = e % n; const long long b = e / n;
            if (P.dx0) P.dx0[e] = P.db ? P.db[b * neq + (T - 1) * n + i] : 0.0;
        }
    }
}

// ---- rollout + cost + backtracking line search, one thread per trajectory ------------------------
// qp_wrapper.MPC.line_search (qp_wrapper.py:417-436) with rollout (:598-611) and compute_cost
// (:690-692): alpha = 1; up to max_iter rounds: u_try = u + alpha du, x_try = rollout(x0, u_try),
// cost_try; a sample whose cost dropped below cost(x, u) keeps its alpha, the others get
// alpha *= decay and try again.  (The reference stops the ROUNDS when every sample improved; a
// sample that already improved is recomputed with the same alpha, so the outcome per sample is
// what this loop gives, including the reference's quirk that a sample that never improves returns
// its last trial together with an alpha decayed once more.)
constexpr int LS_MAXN = 12, LS_MAXM = 8;
struct LsP {
    const double *F, *f, *x0, *x, *u, *du, *C, *c;
    double *xn, *un, *alpha, *cost;
    double decay, dt;
    int B, n, m, T, dyn, max_iter;
};

template <class Map>
__device__ __forceinline__ void ls_step(const double *xs, const double *us, double dt, double *out)
{
    double xa[Map::NX], ua[Map::NU], o[Map::NX];
#pragma unroll
    for (int k = 0; k < Map::NX; ++k) xa[k] = xs[k];
#pragma unroll
    for (int k = 0; k < Map::NU; ++k) ua[k] = us[k];
    Map::template step<double>(xa, ua, dt, o);
#pragma unroll
    for (int k = 0; k < Map::NX; ++k) out[k] = o[k];
}

// Stage cost in the reference's own order of operations (qp_wrapper.py:690-692:
//   0.5 * ((xu[..., None] * C).sum(-2) * xu).sum(-1).sum(-1) + (xu * c).sum(-1).sum(-1)):
// v_j = sum_i tau_i C_ij, quad_t = sum_j v_j tau_j, lin_t = sum_j tau_j c_j, every product rounded before it is added
// (no FMA contraction: torch multiplies, then reduces), the quadratic and the linear part summed over the knots
// separately and combined once (Cost::total).  The line search's accept test compares two costs that are equal up to
// round-off at a converged SQP iterate: with the same association as the reference the tie falls the same way.
struct Cost {
    double quad = 0.0, lin = 0.0;
    __device__ __forceinline__ double total() const { return __dadd_rn(__dmul_rn(0.5, quad), lin); }
};
__device__ __forceinline__ void stage_cost(const LsP &P, long long b, int t, const double *xs, const double *us, int n, int m, Cost &acc)
{
    const int nt = n + m;
    if (!P.C) return;                         // rollout only (dqp_mpc_line_search with C == NULL)
    const double *Ct = P.C + ((long long)t * P.B + b) * nt * nt, *ct = P.c + ((long long)t * P.B + b) * nt;
    double q = 0.0, l = 0.0;
    for (int j = 0; j < nt; ++j) {
        const double tj = j < n ? xs[j] : us[j - n];
        double v = 0.0;
        for (int i = 0; i < nt; ++i) v = __dadd_rn(v, __dmul_rn(i < n ? xs[i] : us[i - n], Ct[i * nt + j]));
        q = __dadd_rn(q, __dmul_rn(v, tj));
        l = __dadd_rn(l, __dmul_rn(tj, ct[j]));
    }
    acc.quad = __dadd_rn(acc.quad, q);
    acc.lin = __dadd_rn(acc.lin, l);
}

// One instantiation per dynamics (the LinDx form with run-time sizes, one per registered model with its
// sizes as constants): a single kernel with a run-time switch over the models takes the registers and
// scratch of the largest of them (the 12-state quadrotor) for every model.
struct LinStep { static constexpr bool LIN = true, FIXED = false; static constexpr int NX = LS_MAXN, NU = LS_MAXM; };
// LinDx at a compile-time size: loops unroll, the state lives in registers instead of scratch
template <int N, int M_> struct LinStepT { static constexpr bool LIN = true, FIXED = true; static constexpr int NX = N, NU = M_; };
template <class Map> struct ModelStep { static constexpr bool LIN = false, FIXED = true; static constexpr int NX = Map::NX, NU = Map::NU; using M = Map; };
#define DQP_LIN_SIZES X(3, 3) X(3, 1) X(4, 1) X(6, 1) X(2, 1) X(4, 2) X(3, 2) X(12, 4)

// the same with the knot's C_t, c_t in registers (same order of operations as stage_cost)
template <int NX, int NU>
__device__ __forceinline__ void stage_cost_regs(const double (&Ct)[(NX + NU) * (NX + NU)], const double (&ct)[NX + NU],
                                                const double (&xs)[NX], const double (&us)[NU], Cost &acc)
{
    constexpr int NT = NX + NU;
    double q = 0.0, l = 0.0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const double tj = j < NX ? xs[j < NX ? j : 0] : us[j >= NX ? j - NX : 0];
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < NT; ++i) v = __dadd_rn(v, __dmul_rn(i < NX ? xs[i < NX ? i : 0] : us[i >= NX ? i - NX : 0], Ct[i * NT + j]));
        q = __dadd_rn(q, __dmul_rn(v, tj));
        l = __dadd_rn(l, __dmul_rn(tj, ct[j]));
    }
    acc.quad = __dadd_rn(acc.quad, q);
    acc.lin = __dadd_rn(acc.lin, l);
}

// Registered models with small knots: the thread is a chain of T knots "load C_t, c_t, u_t, du_t -> cost ->
// model step", one memory round trip per knot and a second pass for the cost of the current trajectory.  Here
// the next knot's inputs are loaded right after the cost of this one, under its model step (atan2 / sincos /
// an RK4 step: long enough to cover the latency), and round 0 evaluates both costs from the one copy of C_t.
template <class S>
__device__ __forceinline__ void line_search_model(const LsP &P, long long b)
{
    constexpr int NX = S::NX, NU = S::NU, NT = NX + NU;
    const double *__restrict__ pC = P.C, *__restrict__ pc = P.c, *__restrict__ pu = P.u, *__restrict__ pdu = P.du;
    const double *__restrict__ px = P.x, *__restrict__ px0 = P.x0;
    double *__restrict__ pxn = P.xn, *__restrict__ pun = P.un;
    const int T = P.T;
    const long long B = P.B;
    double Ct[NT * NT], ct[NT], u0[NU], du[NU], xc[NX];
    auto load = [&](int t, bool current) {
        const long long k = (long long)t * B + b;
#pragma unroll
        for (int i = 0; i < NT * NT; ++i) Ct[i] = pC[k * (NT * NT) + i];
#pragma unroll
        for (int i = 0; i < NT; ++i) ct[i] = pc[k * NT + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) { u0[i] = pu[k * NU + i]; du[i] = pdu ? pdu[k * NU + i] : 0.0; }
        if (current) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = px[k * NX + i];
        }
    };
    double alpha = 1.0, cost_try = 0.0, cost_here = 0.0;
    Cost here;
    for (int round = 0; round < P.max_iter; ++round) {
        double xs[NX], us[NU], nx[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[i] = px0[b * NX + i];
        Cost tr;
        load(0, round == 0);
        for (int t = 0; t < T; ++t) {
            const long long k = (long long)t * B + b;
#pragma unroll
            for (int i = 0; i < NU; ++i) { us[i] = pdu ? u0[i] + du[i] * alpha : u0[i]; pun[k * NU + i] = us[i]; }
#pragma unroll
            for (int i = 0; i < NX; ++i) pxn[k * NX + i] = xs[i];
            if (round == 0) stage_cost_regs<NX, NU>(Ct, ct, xc, u0, here);
            stage_cost_regs<NX, NU>(Ct, ct, xs, us, tr);
            if (t == T - 1) break;
            load(t + 1, round == 0);
            ls_step<typename S::M>(xs, us, P.dt, nx);
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = nx[i];
        }
        cost_try = tr.total();
        if (round == 0) cost_here = here.total();
        if (cost_try < cost_here) break;            // improved: this alpha stands
        alpha *= P.decay;                           // qp_wrapper.py:431-432
    }
    P.alpha[b] = alpha;
    P.cost[b] = cost_try;
}

template <class S>
__global__ __launch_bounds__(64) void line_search_kernel(LsP P)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    if constexpr (!S::LIN && (S::NX + S::NU) * (S::NX + S::NU) <= 64) {
        if (P.C) { line_search_model<S>(P, b); return; }
    }
    const int n = S::FIXED ? S::NX : P.n, m = S::FIXED ? S::NU : P.m, T = P.T, nt = n + m;
    Cost here;
    for (int t = 0; t < T && P.C; ++t) {
        double xs[S::NX], us[S::NU];
        for (int i = 0; i < n; ++i) xs[i] = P.x[((long long)t * P.B + b) * n + i];
        for (int i = 0; i < m; ++i) us[i] = P.u[((long long)t * P.B + b) * m + i];
        stage_cost(P, b, t, xs, us, n, m, here);
    }
    const double cost_here = here.total();
    double alpha = 1.0, cost_try = 0.0;
    for (int round = 0; round < P.max_iter; ++round) {
        double xs[S::NX], us[S::NU], nx[S::NX];
        for (int i = 0; i < n; ++i) xs[i] = P.x0[b * n + i];
        Cost tr;
        for (int t = 0; t < T; ++t) {
            for (int i = 0; i < m; ++i) {
                const long long o = ((long long)t * P.B + b) * m + i;
                us[i] = P.du ? P.u[o] + P.du[o] * alpha : P.u[o];
                P.un[o] = us[i];
            }
            for (int i = 0; i < n; ++i) P.xn[((long long)t * P.B + b) * n + i] = xs[i];
            stage_cost(P, b, t, xs, us, n, m, tr);
            if (t == T - 1) break;
            if constexpr (S::LIN) {     // LinDx: x+ = F_t [x; u] + f_t
                const double *Ft = P.F + ((long long)t * P.B + b) * n * nt, *ft = P.f + ((long long)t * P.B + b) * n;
                for (int i = 0; i < n; ++i) {
                    double a = ft[i];
                    for (int j = 0; j < nt; ++j) a += Ft[i * nt + j] * (j < n ? xs[j] : us[j - n]);
                    nx[i] = a;
                }
            } else {
                ls_step<typename S::M>(xs, us, P.dt, nx);
            }
            for (int i = 0; i < n; ++i) xs[i] = nx[i];
        }
        cost_try = tr.total();
        if (cost_try < cost_here) break;            // improved: this alpha stands
        alpha *= P.decay;                           // qp_wrapper.py:431-432
    }
    P.alpha[b] = alpha;
    P.cost[b] = cost_try;
}

// ---- adjoint of the rollout (what autograd derives from qp_wrapper.py:598-611), one thread per
// trajectory: lam_{T-1} = g_{T-1};  for t = T-2 .. 0:  du_t = Ju_t^T lam_{t+1},
// lam_t = g_t + Jx_t^T lam_{t+1};  dF_t = lam_{t+1} [x_t; u_t]^T, df_t = lam_{t+1} (LinDx);  dx0 = lam_0.
struct RbP {
    const double *F, *x, *u, *g;
    double *dx0, *du, *dF, *df;
    double dt;
    int B, n, m, T, dyn;
};

template <class Map>
__device__ __forceinline__ void vjp_step(const double *xs, const double *us, double dt, const double *lam,
                                         double *gx, double *gu)
{
    // forward-mode seeds in chunks of KC directions (all at once for the small models; a 16-seed
    // dual of the 12-state RK4 step would not fit the register file)
    constexpr int NX = Map::NX, NU = Map::NU, K = NX + NU, KC = K <= 8 ? K : 4;
    using S = dqp::dyn::Dual<KC>;
#pragma unroll
    for (int c0 = 0; c0 < K; c0 += KC) {
        S xa[NX], ua[NU], o[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) { xa[k] = S(xs[k]); if (k >= c0 && k < c0 + KC) xa[k].d[k - c0] = 1.0; }
#pragma unroll
        for (int k = 0; k < NU; ++k) { ua[k] = S(us[k]); if (NX + k >= c0 && NX + k < c0 + KC) ua[k].d[NX + k - c0] = 1.0; }
        Map::template step<S>(xa, ua, dt, o);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int col = c0 + c;
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < NX; ++r) a += o[r].d[c] * lam[r];
            if (col < NX) gx[col] = a;
            else if (col < K) gu[col - NX] = a;
        }
    }
}

template <class S>
__global__ __launch_bounds__(64) void rollout_backward_kernel(RbP P)
{
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    const int n = S::FIXED ? S::NX : P.n, m = S::FIXED ? S::NU : P.m, T = P.T, nt = n + m;
    double lam[S::NX], gx[S::NX], gu[S::NU], xs[S::NX], us[S::NU];
    for (int i = 0; i < n; ++i) lam[i] = P.g[((long long)(T - 1) * P.B + b) * n + i];
    if (P.du) for (int i = 0; i < m; ++i) P.du[((long long)(T - 1) * P.B + b) * m + i] = 0.0;   // last action unused
    for (int t = T - 2; t >= 0; --t) {
        for (int i = 0; i < n; ++i) xs[i] = P.x[((long long)t * P.B + b) * n + i];
        for (int i = 0; i < m; ++i) us[i] = P.u[((long long)t * P.B + b) * m + i];
        if constexpr (S::LIN) {
            const double *Ft = P.F + ((long long)t * P.B + b) * n * nt;
            for (int j = 0; j < nt; ++j) {
                double a = 0.0;
                for (int i = 0; i < n; ++i) a += Ft[i * nt + j] * lam[i];
                if (j < n) gx[j] = a; else gu[j - n] = a;
            }
            if (P.dF) {
                double *dFt = P.dF + ((long long)t * P.B + b) * n * nt;
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < nt; ++j) dFt[i * nt + j] = lam[i] * (j < n ? xs[j] : us[j - n]);
            }
            if (P.df) for (int i = 0; i < n; ++i) P.df[((long long)t * P.B + b) * n + i] = lam[i];
        } else {
            vjp_step<typename S::M>(xs, us, P.dt, lam, gx, gu);
        }
        if (P.du) for (int i = 0; i < m; ++i) P.du[((long long)t * P.B + b) * m + i] = gu[i];
        for (int i = 0; i < n; ++i) lam[i] = P.g[((long long)t * P.B + b) * n + i] + gx[i];
    }
    if (P.dx0) for (int i = 0; i < n; ++i) P.dx0[b * n + i] = lam[i];
}

int check(const dqp_mpc_dims *d)
{
    if (!d || d->nbatch < 0 || d->n_state <= 0 || d->n_ctrl <= 0 || d->T < 2) return DQP_ERR_BAD_ARG;
    return DQP_OK;
}

int grid_for(long long total)
{
    long long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));     // grid-stride beyond 8 blocks per CU
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int
dqp_mpc_assemble(const dqp_mpc_dims *d, const double *C, const double *c, const double *F,
                 const double *f, const double *x0, const double *u_lower, const double *u_upper,
                 double *Q, double *p, double *G, double *h, double *A, double *b, void *stream)
{
    int rc = check(d);
    if (rc) return rc;
    if (d->nbatch == 0) return DQP_OK;
    if (!C || !c || !F || !f || !x0 || !Q || !p || !G || !h || !A || !b) return DQP_ERR_BAD_ARG;
    if (d->has_bounds && (!u_lower || !u_upper)) return DQP_ERR_BAD_ARG;
    MpcP P = {};
    P.C = C; P.c = c; P.F = F; P.f = f; P.x0 = x0; P.ul = u_lower; P.uu = u_upper;
    P.Q = Q; P.p = p; P.G = G; P.h = h; P.A = A; P.b = b;
    P.B = d->nbatch; P.n = d->n_state; P.m = d->n_ctrl; P.T = d->T; P.bounds = d->has_bounds;
    const long long nt = P.n + P.m, nz = P.T * nt, neq = (long long)P.T * P.n;
    const long long nineq = P.bounds ? 2LL * P.T * P.m : P.m;
    const long long total = (long long)P.B * (nz * nz + neq * nz + nineq * nz + nz + neq + nineq);
    DQP_LAUNCH(assemble_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

__attribute__((visibility("default"))) int
dqp_mpc_assemble_backward(const dqp_mpc_dims *d, const double *dQ, const double *dp,
                          const double *dA, const double *db, double *dC, double *dc, double *dF,
                          double *df, double *dx0, void *stream)
{
    int rc = check(d);
    if (rc) return rc;
    if (d->nbatch == 0) return DQP_OK;
    MpcP P = {};
    P.dQ = dQ; P.dp = dp; P.dA = dA; P.db = db;
    P.dC = dC; P.dc = dc; P.dF = dF; P.df = df; P.dx0 = dx0;
    P.B = d->nbatch; P.n = d->n_state; P.m = d->n_ctrl; P.T = d->T; P.bounds = d->has_bounds;
    const long long nt = P.n + P.m;
    const long long total = (long long)P.B * (P.T * nt * nt + P.T * nt + (P.T - 1) * P.n * nt +
                                              (P.T - 1) * P.n + P.n);
    DQP_LAUNCH(assemble_backward_kernel, dim3(grid_for(total)), dim3(256), 0,
                       (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

__attribute__((visibility("default"))) int
dqp_mpc_line_search(const dqp_mpc_dims *d, int dyn_id, double dt, const double *F, const double *f,
                    const double *x0, const double *x, const double *u, const double *delta_u,
                    const double *C, const double *c, double decay, int32_t max_iter, double *x_new,
                    double *u_new, double *alpha, double *cost_new, void *stream)
{
    int rc = check(d);
    if (rc) return rc;
    if (d->nbatch == 0) return DQP_OK;
    if (d->n_state > LS_MAXN || d->n_ctrl > LS_MAXM) return DQP_ERR_TOO_LARGE;
    // C == NULL: plain rollout of u (delta_u, x, c may be NULL too; one round, cost reported as 0)
    if (!x0 || !u || !x_new || !u_new || !alpha || !cost_new || max_iter < 1) return DQP_ERR_BAD_ARG;
    if (C && (!x || !delta_u || !c)) return DQP_ERR_BAD_ARG;
    if (!C) max_iter = 1;
    if (dyn_id == 0) {
        if (!F || !f) return DQP_ERR_BAD_ARG;
    } else {
        int32_t n = 0, m = 0;
        if (dqp_dyn_sizes(dyn_id, &n, &m) != DQP_OK || n != d->n_state || m != d->n_ctrl) return DQP_ERR_BAD_ARG;
    }
    LsP P = {};
    P.F = F; P.f = f; P.x0 = x0; P.x = x; P.u = u; P.du = delta_u; P.C = C; P.c = c;
    P.xn = x_new; P.un = u_new; P.alpha = alpha; P.cost = cost_new;
    P.decay = decay; P.dt = dt; P.B = d->nbatch; P.n = d->n_state; P.m = d->n_ctrl; P.T = d->T;
    P.dyn = dyn_id; P.max_iter = max_iter;
    const dim3 grid((P.B + 63) / 64), block(64);
    hipStream_t st = (hipStream_t)stream;
    using namespace dqp::dyn;
    switch (dyn_id) {
    case 0:
#define X(a, b) if (P.n == a && P.m == b) { DQP_LAUNCH((line_search_kernel<LinStepT<a, b>>), grid, block, 0, st, P); break; }
        DQP_LIN_SIZES
#undef X
        DQP_LAUNCH(line_search_kernel<LinStep>, grid, block, 0, st, P); break;
    case DQP_DYN_PENDULUM1L: DQP_LAUNCH(line_search_kernel<ModelStep<Robot<Pendulum1l>>>, grid, block, 0, st, P); break;
    case DQP_DYN_CARTPOLE1L: DQP_LAUNCH(line_search_kernel<ModelStep<Robot<Cartpole1l>>>, grid, block, 0, st, P); break;
    case DQP_DYN_CARTPOLE2L: DQP_LAUNCH(line_search_kernel<ModelStep<Robot<Cartpole2l>>>, grid, block, 0, st, P); break;
    case DQP_DYN_PENDULUM_EULER: DQP_LAUNCH(line_search_kernel<ModelStep<PendulumEuler>>, grid, block, 0, st, P); break;
    case DQP_DYN_PENDULUM_DX: DQP_LAUNCH(line_search_kernel<ModelStep<PendulumDx>>, grid, block, 0, st, P); break;
    case DQP_DYN_REXQUADROTOR: DQP_LAUNCH(line_search_kernel<ModelStep<RexQuadrotor>>, grid, block, 0, st, P); break;
    default: return DQP_ERR_BAD_ARG;
    }
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

__attribute__((visibility("default"))) int
dqp_mpc_rollout_backward(const dqp_mpc_dims *d, int dyn_id, double dt, const double *F, const double *x,
                         const double *u, const double *g_x, double *d_x0, double *d_u, double *d_F,
                         double *d_f, void *stream)
{
    int rc = check(d);
    if (rc) return rc;
    if (d->nbatch == 0) return DQP_OK;
    if (d->n_state > LS_MAXN || d->n_ctrl > LS_MAXM) return DQP_ERR_TOO_LARGE;
    if (!x || !u || !g_x) return DQP_ERR_BAD_ARG;
    if (dyn_id == 0) {
        if (!F) return DQP_ERR_BAD_ARG;
    } else {
        int32_t n = 0, m = 0;
        if (dqp_dyn_sizes(dyn_id, &n, &m) != DQP_OK || n != d->n_state || m != d->n_ctrl) return DQP_ERR_BAD_ARG;
    }
    RbP P = {};
    P.F = F; P.x = x; P.u = u; P.g = g_x; P.dx0 = d_x0; P.du = d_u; P.dF = d_F; P.df = d_f;
    P.dt = dt; P.B = d->nbatch; P.n = d->n_state; P.m = d->n_ctrl; P.T = d->T; P.dyn = dyn_id;
    const dim3 grid((P.B + 63) / 64), block(64);
    hipStream_t st = (hipStream_t)stream;
    using namespace dqp::dyn;
    switch (dyn_id) {
    case 0:
#define X(a, b) if (P.n == a && P.m == b) { DQP_LAUNCH((rollout_backward_kernel<LinStepT<a, b>>), grid, block, 0, st, P); break; }
        DQP_LIN_SIZES
#undef X
        DQP_LAUNCH(rollout_backward_kernel<LinStep>, grid, block, 0, st, P); break;
    case DQP_DYN_PENDULUM1L: DQP_LAUNCH(rollout_backward_kernel<ModelStep<Robot<Pendulum1l>>>, grid, block, 0, st, P); break;
    case DQP_DYN_CARTPOLE1L: DQP_LAUNCH(rollout_backward_kernel<ModelStep<Robot<Cartpole1l>>>, grid, block, 0, st, P); break;
    case DQP_DYN_CARTPOLE2L: DQP_LAUNCH(rollout_backward_kernel<ModelStep<Robot<Cartpole2l>>>, grid, block, 0, st, P); break;
    case DQP_DYN_PENDULUM_EULER: DQP_LAUNCH(rollout_backward_kernel<ModelStep<PendulumEuler>>, grid, block, 0, st, P); break;
    case DQP_DYN_PENDULUM_DX: DQP_LAUNCH(rollout_backward_kernel<ModelStep<PendulumDx>>, grid, block, 0, st, P); break;
    case DQP_DYN_REXQUADROTOR: DQP_LAUNCH(rollout_backward_kernel<ModelStep<RexQuadrotor>>, grid, block, 0, st, P); break;
    default: return DQP_ERR_BAD_ARG;
    }
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // extern "C"
