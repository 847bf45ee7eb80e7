// dqp_al_banded.hip -- NewtonAL for registered device models with the MPC structure exploited.
//
// The augmented-Lagrangian Hessian of the reference (qpth/al_utils.py:62-102: diag(Q) + rho Jc^T Jc
// with Jc the clamped constraint Jacobian of al_utils.py:162-318) is block tridiagonal in the
// knots: a dynamics row block couples knot t and t+1 only, box rows touch one control.  The
// reference builds the dense (B, ncon, nz) Jacobian, a dense bmm and a dense Cholesky
// (O(nz^3), nz = T (n+m)); here one launch per Newton step does, per problem,
//
//   forward sweep over the knots t = 0 .. T-1  (everything of a knot lives in registers):
//     J_t = [df/dx, df/du](x_t, u_t)      lane i evaluates the model with ONE forward-mode seed e_i
//                                         -> column i of J_t: 16 lanes = all columns at once
//     g_t  = Q z + q + J^T (lam + rho res_c)                             (merit gradient rows)
//     H_tt = diag(Q) + rho (J_t^T J_t + I_x + active box indicators),  H_{t+1,t} = -rho E_x^T J_t
//     S_t  = H_tt - L_{t,t-1} L_{t,t-1}^T ;  L_tt = chol(S_t) ;  L_{t+1,t}^T = L_tt^-1 H_{t+1,t}^T
//     y_t  = L_tt^-1 (-g_t - L_{t,t-1} y_{t-1})
//   backward sweep t = T-1 .. 0:   upd_t = L_tt^-T (y_t - L_{t+1,t}^T upd_{t+1})
//
// i.e. the block Cholesky of the same matrix: O(T (n+m)^3) instead of O(T^3 (n+m)^3), no Jacobian or
// Hessian in HBM.  One problem per 16-lane DPP row (4 per wavefront), matrices row-distributed over
// the lanes as in the QP kernels (dqp_r16_prims.h).  The factor is kept in its banded form
// (per knot: L_tt rows, 1/diag, L_{t+1,t}^T rows) for the backward pass (al_utils.py:477-480).
//
// Variants of the sweep (all the same arithmetic up to summation order): the substitutions run on the unit-triangular
// form (unit_lower / trsv_unit: one broadcast + one fma per step), M^T M and M^T y come from an LDS-transposed copy of M,
// L^-T is applied in broadcast form on the lane's column of L (trsvT_bcast); small models (nt <= 8) prefetch a knot ahead,
// evaluate the model of G / nt knots at once on the group's idle lanes (KB) and, for short horizons, keep the factor in
// LDS between the sweeps (LF: it is written to `fac` only from the step whose factor the caller keeps).
//
// HBM per Newton step and problem: xu, Qd, q (3 nz) + lam (ncon) in, update (nz) + banded factor
// (T (nt^2 + nt n + nt)) out -- e.g. 6.6 KB at cartpole T = 20 against 177 KB for the dense path
// (Jc written + read, L written).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dqp.h"
#include "dqp_r16_prims.h"
#include "dqp_dyn_models.h"

namespace {

using namespace dqp::r16;
using dqp::dyn::Dual;

struct BandP {
    const double *xu, *x0, *Qd, *q, *lam, *rho, *ul, *uu;    // forward inputs
    const double *rhs;                                       // solve-only mode: right-hand side (B, nz)
    const double *xnext, *Jx, *Ju;                           // caller-supplied linearisation (Given<>): f(x_t,u_t) (B,T-1,n),
                                                             // df/dx (B,T-1,n,n), df/du (B,T-1,n,m)
    double *upd;        // (B, T, nt): -H^-1 grad  (or -H^-1 rhs in solve-only mode)
    double *fac;        // banded factor, per (b, t): nt rows x (nt + 1 + nx) doubles
    int32_t *info;      // (B): 0 or 1 + knot of the first non-positive pivot
    double dt;
    int B, T;
    int keep;           // 0: the caller does not need this step's factor (an intermediate Newton step) -- kernels that hold the
                        // factor on chip (LF below) then leave `fac` untouched; 1: write it
#ifdef DQP_BAND_STAMPS
    unsigned long long *stamps;     // instrumented build (tools/stamps_band.py): 8 accumulated s_memtime phases per workgroup
#endif
};

// Phase clock of the instrumented build (tools/stamps_band.py; compiled out otherwise).  Each stamp drains the memory
// counters first, so the phases do not overlap as they do in the shipped kernel: the split shows where the cycles of a
// knot go, not what removing a phase would save.
#ifdef DQP_BAND_STAMPS
static unsigned long long *g_band_stamps = nullptr;
#define BAND_STAMP(k)                                                                          \
    do {                                                                                       \
        unsigned long long now_;                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
        stamp_acc[k] += now_ - stamp_last; stamp_last = now_;                                  \
    } while (0)
#define BAND_STAMP_INIT                                                                        \
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory")
#else
#define BAND_STAMP(k)
#define BAND_STAMP_INIT
#endif

// A dynamics the caller linearised itself (a torch module with its own Jacobians, deqmpc/envs.py:50-82,
// rex_quadrotor.py:131-146): sizes only; the kernel reads f, df/dx, df/du from memory instead of evaluating a
// registered model with forward-mode seeds.
template <int NX_, int NU_> struct Given { static constexpr int NX = NX_, NU = NU_; };
template <class M> struct is_given { static constexpr bool value = false; };
template <int A, int B_> struct is_given<Given<A, B_>> { static constexpr bool value = true; };

template <class Map> struct BandCfg {
    static constexpr int NX = Map::NX, NU = Map::NU, NT = NX + NU;
    // per knot: ROW = nt + 1 + nx "columns" (L row entries, 1/diag, M row entries) of nt doubles each, one per lane of the
    // knot: element c of lane r's row at [c nt + r], so that one store / load instruction moves nt consecutive doubles per
    // problem.  (Lane-major rows -- [r ROW + c] -- made every instruction touch 64 separate cache lines: the factor is
    // 111 KB per quadrotor problem, written by the forward sweep and read by the backward sweep of every Newton step.)
    static constexpr int ROW = NT + 1 + NX;
    static_assert(NT <= 16, "one knot must fit a 16-lane DPP row");
};

// ---- lane groups.  A knot of nt <= 8 rows uses half of a 16-lane DPP row; the models this solver is run on
// in the reference's experiments (pendulum nt 3, cartpole-1 nt 5, cartpole-2 nt 7) are all of that kind, and two
// thirds of the kernel is the per-lane forward-mode model evaluation, which does not care how wide a group is.
// Grp<8> puts one problem on each HALF row (8 per wavefront): a broadcast is two bank-masked row_newbcast moves
// (lanes 0-7 take lane k, lanes 8-15 take lane 8 + k) instead of one, a group sum three DPP steps instead of four.
template <int G> struct Grp;
template <> struct Grp<16> {
    __device__ __forceinline__ static double rb(double v, int k) { return dqp::r16::rb(v, k); }
    __device__ __forceinline__ static double sum(double v) { return row_sum(v); }
};
template <int K> __device__ __forceinline__ double rb8k(double v)
{
    const double lo = __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0x3, false);       // banks 0,1: lanes 0-7 of the row
    return __builtin_amdgcn_update_dpp(lo, v, 0x158 + K, 0xf, 0xc, false);                  // banks 2,3: lanes 8-15
}
template <> struct Grp<8> {
    __device__ __forceinline__ static double rb(double v, int k)
    {
        switch (k & 7) {
        case 0: return rb8k<0>(v);  case 1: return rb8k<1>(v);  case 2: return rb8k<2>(v);  case 3: return rb8k<3>(v);
        case 4: return rb8k<4>(v);  case 5: return rb8k<5>(v);  case 6: return rb8k<6>(v);  default: return rb8k<7>(v);
        }
    }
    __device__ __forceinline__ static double sum(double v)
    {
        v += dppd<0x141>(v);        // row_half_mirror: l <-> 7 - l
        v += dppd<0xb1>(v);         // quad_perm [1,0,3,2]
        v += dppd<0x4e>(v);         // quad_perm [2,3,0,1]
        return v;
    }
};

// Cholesky / triangular solves of one row-distributed N x N block per group (dqp_r16_prims.h's chol_rows /
// trsv_rows at S = 1, on the group's broadcast)
template <int G, int N>
__device__ __forceinline__ bool chol_g(double (&L)[1][N], double (&rd)[1], int r)
{
    if constexpr (G == 16) return chol_rows<1, N>(L, rd, r);
    else {
        bool ok = true;
        rd[0] = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double dk = Grp<G>::rb(L[0][k], k);
            if (!(dk > 0.0)) { ok = false; dk = 1.0; }
            const double ri = frsqrt(dk);
            if (r == k) rd[0] = ri;
            PIN(rd[0]);
            const double col = (r >= k) ? L[0][k] * ri : 0.0;
            L[0][k] = col;
#pragma unroll
            for (int j = k + 1; j < N; ++j) L[0][j] = fma(-col, Grp<G>::rb(col, j), L[0][j]);
        }
        return ok;
    }
}
template <int G, int N>
__device__ __forceinline__ void trsv_g(const double (&L)[1][N], const double (&rd)[1], double (&b)[1], int r)
{
    if constexpr (G == 16) trsv_rows<1, N>(L, rd, b, r);
    else {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double yk = Grp<G>::rb(b[0] * rd[0], k);
            b[0] = (r == k) ? yk : fma(r > k ? -L[0][k] : 0.0, yk, b[0]);
        }
    }
}
// The forward sweep solves nx + 1 systems with each knot's L_tt.  With L = Lu D (Lu unit lower triangular, D = diag L)
// the substitution runs on w = D y:  w_k = b_k - sum_{j<k} (L[k][j] / L[j][j]) w_j  -- per step one broadcast and one
// fma, no scaling and no lane selects (trsv_g above: a multiply and two selects per step and system); y = w rd at the
// end.  unit_lower overwrites the lane's row of L by its row of Lu with a ZERO diagonal (rows of lanes beyond the
// matrix are identity rows and become zero rows).
template <int G, int N>
__device__ __forceinline__ void unit_lower(double (&L)[1][N], const double (&rd)[1], int r)
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double s = Grp<G>::rb(rd[0], k);
        L[0][k] = (r > k) ? L[0][k] * s : 0.0;
    }
}
template <int G, int N>
__device__ __forceinline__ double trsv_unit(const double (&Lu)[1][N], double b)
{
#pragma unroll
    for (int k = 0; k < N - 1; ++k) b = fma(-Lu[0][k], Grp<G>::rb(b, k), b);
    return b;           // w; the caller scales by rd
}

// b <- L^-T b for a row-distributed lower-triangular NT x NT matrix in registers
template <int G, int NT>
__device__ __forceinline__ void trsvT_rows(const double (&L)[1][NT], const double (&rd)[1], double (&b)[1], int r)
{
#pragma unroll
    for (int j = NT - 1; j >= 0; --j) {
        const double part = (r > j && r < NT) ? L[0][j] * b[0] : 0.0;
        const double tot = Grp<G>::sum(part);
        if (r == j) b[0] = (b[0] - tot) * rd[0];
    }
}

// b <- L^-T b by broadcasts: with the lane holding its COLUMN of L (Lt[k] = L[k][r]: row r of the upper triangular L^T)
// the substitution U x = b runs like the forward one -- w = D x: w_r = b_r - sum_{k>r} L[k][r] / L[k][k] w_k, one
// broadcast and one fma per step on a chain of 2 dependent instructions, instead of a group sum (a multiply and four
// DPP + add rounds in sequence) per step.  The column comes from the row through the group's LDS tile `tile` (element
// (row, col) at [col TS + row], as the forward sweep's transposed copy of M).
template <int G, int NT, int TS>
__device__ __forceinline__ double trsvT_bcast(const double (&L)[1][NT], double rdv, double b, double *trow, const double *tcol, int r)
{
#pragma unroll
    for (int c = 0; c < NT; ++c) trow[c * TS] = L[0][c];              // L[r][c] -> tile (row r, col c)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double U[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) U[k] = tcol[k];                      // column r: L[k][r]
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 1; k < NT; ++k) {
        const double sk = Grp<G>::rb(rdv, k);         // outside the select: a broadcast inside a divergent arm would read
        U[k] = (r < k) ? U[k] * sk : 0.0;             // lane k while lane k is masked off (DPP returns 0 for it)
    }
#pragma unroll
    for (int k = NT - 1; k >= 1; --k) b = fma(-U[k], Grp<G>::rb(b, k), b);
    return b * rdv;
}

// LF ("LDS factor", short horizons of the 6- to 8-row models: config 5's cartpole-2 at T = 5): the factor rows never
// leave the chip between the forward and the backward sweep -- element k of knot t of lane (group p, row r) at
// lf[(t (ROW + 1) + k) NL + p nt + r], NL = groups x nt (the idle lanes of a group take no space) -- and go to `fac` only
// when the caller keeps this step's factor (BandP.keep: the last Newton step of a solve, which NewtonAL.backward
// uses).  At B = 65536 the factor round trip (3.9 KB written and read back per problem and step, against 1 KB of inputs)
// was most of the kernel's HBM traffic, in phase over all wavefronts.
template <class Map, int G, bool LF = false>
__global__ __launch_bounds__(64) void al_banded_newton_kernel(BandP P)
{
    using C = BandCfg<Map>;
    using Gr = Grp<G>;
    constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    static_assert(NT <= G, "a knot must fit its lane group");
    const int lane = threadIdx.x, r = lane & (G - 1);
    long long b = (long long)blockIdx.x * (64 / G) + lane / G;
    const bool live = b < P.B;
    if (!live) b = P.B - 1;
    const int T = P.T, neq = T * NX;
    const double *xu = P.xu + b * (long long)T * NT, *Qd = P.Qd + b * (long long)T * NT, *q = P.q + b * (long long)T * NT;
    const double *lam = P.lam + b * (long long)(neq + 2 * T * NU), *x0 = P.x0 + b * NX;
    const double rho = P.rho[b];
    double *fac = P.fac + b * (long long)T * NT * C::ROW;
    const bool inT = r < NT;

    double Mprev[NX], yprev[1] = {0.0}, mu_prev[NX];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < NX; ++j) { Mprev[j] = 0.0; mu_prev[j] = 0.0; }

    // the group's LDS tile (element (row, col) at [col TS + row]): the forward sweep's transposed copy of M_t, then the
    // backward sweep's transposed L_tt.  TS = G + 2 doubles puts a group's 16-byte column reads on distinct bank quads; PS
    // shifts neighbouring 16-lane groups by half the banks for the row-wise writes.
    constexpr int TS = G + 2, PS = G * TS + (G == 16 ? 16 : 0);
    __shared__ __attribute__((aligned(16))) double trs[(64 / G) * PS];
    double *trow = trs + (lane / G) * PS + r;
    const double *tcol = trs + (lane / G) * PS + r * TS;
    constexpr bool PRE = NT <= 8;        // small models: prefetched knot inputs, delayed factor stores (below)
    static_assert(!LF || PRE, "the LDS-resident factor is for the small models");
    extern __shared__ double lf_dyn[];
    constexpr int NL = (64 / G) * NT, LROW = C::ROW + 1;
    double *lf = lf_dyn + (lane / G) * NT + (inT ? r : 0);          // + (t LROW + k) NL
    BAND_STAMP_INIT;
    if constexpr (!PRE) {
        static_assert(G == 16, "the large-model sweep keeps one knot per DPP row");
        // transposed copy of M_t (nt rows over the lanes, nx columns in registers) in the tile: columns nx .. 15 stay zero,
        // lanes beyond nx read them and so subtract nothing
    #pragma unroll
        for (int j = NX; j < G; ++j) trow[j * TS] = 0.0;
        // the knot and its successor's state (uniform loads), fetched one knot ahead: vmcnt is in order on this ISA, so
        // loads issued behind a knot's nt + 1 + nx factor stores wait for those stores too -- the next knot's are issued in
        // front of them (they are the model's inputs: nothing else has to stay live for it)
        double pz[NT], pxn1[NX];
        auto load_state = [&](int t) {
    #pragma unroll
            for (int j = 0; j < NT; ++j) pz[j] = xu[t * NT + j];
    #pragma unroll
            for (int j = 0; j < NX; ++j) pxn1[j] = (t < T - 1) ? xu[(t + 1) * NT + j] : 0.0;
        };
        load_state(0);
        for (int t = 0; t < T; ++t) {
            double z[NT], xn1[NX];
    #pragma unroll
            for (int j = 0; j < NT; ++j) z[j] = pz[j];
            const bool dynrow = t < T - 1;
    #pragma unroll
            for (int j = 0; j < NX; ++j) xn1[j] = pxn1[j];
            BAND_STAMP(0);          // knot loads
            // ---- column r of J_t = [df/dx df/du] by one forward-mode seed per lane
            double Jc[NX], mu[NX];
            if constexpr (is_given<Map>::value) {
                const int tc = dynrow ? t : 0;
                const double *fx = P.xnext + (b * (long long)(T - 1) + tc) * NX;
                const double *jx = P.Jx + (b * (long long)(T - 1) + tc) * NX * NX;
                const double *ju = P.Ju + (b * (long long)(T - 1) + tc) * NX * NU;
    #pragma unroll
                for (int j = 0; j < NX; ++j) {
                    const double col = r < NX ? jx[j * NX + r] : ju[j * NU + (inT ? r - NX : 0)];
                    Jc[j] = (dynrow && inT) ? col : 0.0;
                    const double res = dynrow ? xn1[j] - fx[j] : 0.0;
                    mu[j] = dynrow ? lam[t * NX + j] + rho * res : 0.0;
                }
            } else {
                Dual<1> xs[NX], us[NU], out[NX];
    #pragma unroll
                for (int j = 0; j < NX; ++j) { xs[j] = Dual<1>(z[j]); xs[j].d[0] = (r == j) ? 1.0 : 0.0; }
    #pragma unroll
                for (int j = 0; j < NU; ++j) { us[j] = Dual<1>(z[NX + j]); us[j].d[0] = (r == NX + j) ? 1.0 : 0.0; }
                Map::template step<Dual<1>>(xs, us, P.dt, out);
    #pragma unroll
                for (int j = 0; j < NX; ++j) {
                    Jc[j] = (dynrow && inT) ? out[j].d[0] : 0.0;
                    const double res = dynrow ? xn1[j] - out[j].v : 0.0;                 // x_{t+1} - f(x_t, u_t)
                    mu[j] = dynrow ? lam[t * NX + j] + rho * res : 0.0;
                }
            }
            BAND_STAMP(1);          // model + Jacobian column
            // ---- gradient element r of this knot, and the diagonal terms of H_tt
            // (per-lane addresses: one coalesced load each instead of NT predicated ones)
            const int rr = inT ? r : 0;
            const double zr = xu[t * NT + rr], qdr = Qd[t * NT + rr], qr = q[t * NT + rr];
            double g = qdr * zr + qr, dg = qdr;
    #pragma unroll
            for (int j = 0; j < NX; ++j) g -= Jc[j] * mu[j];
            {
                // x rows: + I block of the previous dynamics rows / the x_0 rows; u rows: the box rows
                const int rx = r < NX ? r : 0, i = (inT && r >= NX) ? r - NX : 0;
                double prev = 0.0;
    #pragma unroll
                for (int j = 0; j < NX; ++j) prev = (r == j) ? mu_prev[j] : prev;
                const double first = lam[(T - 1) * NX + rx] + rho * (zr - x0[rx]);
                const int row = neq + t * 2 * NU + i;
                const double rup = zr - P.uu[i], rlo = P.ul[i] - zr;
                const double lup = lam[row], llo = lam[row + NU];
                if (r < NX) {
                    g += (t > 0) ? prev : first;
                    dg += rho;
                } else if (inT) {
                    g += (lup + rho * fmax(rup, 0.0)) - (llo + rho * fmax(rlo, 0.0));
                    dg += rho * ((rup > 0.0 ? 1.0 : 0.0) + (rlo > 0.0 ? 1.0 : 0.0));
                }
            }
            // ---- H_tt row r:  rho J^T J + diag - Gram(M_prev) on the x-x block
            double H[1][NT], rd[1];
    #pragma unroll
            for (int c = 0; c < NT; ++c) {
                double a = 0.0;
    #pragma unroll
                for (int j = 0; j < NX; ++j) a = fma(Jc[j], Gr::rb(Jc[j], c), a);
                H[0][c] = rho * a + ((r == c) ? dg : 0.0);
            }
            // row r of M_prev^T M_prev and of M_prev^T y_prev from the transposed copy: sum_k M[k][r] M[k][j] with the
            // left factor in the lane's own registers -- one broadcast + one fma per term, no group sums and no selects
            // (12 x 13 / 2 group sums of 4 DPP steps each were a quarter of the knot)
            double Mt[G];
            if (t > 0) {
    #pragma unroll
                for (int k = 0; k < G; ++k) Mt[k] = tcol[k];
    #pragma unroll
                for (int j = 0; j < NX; ++j) {
                    double a = 0.0;
    #pragma unroll
                    for (int k = 0; k < NT; ++k) a = fma(Mt[k], Gr::rb(Mprev[j], k), a);
                    H[0][j] -= a;
                }
            }
    #pragma unroll
            for (int c = 0; c < NT; ++c) H[0][c] = inT ? H[0][c] : ((r == c) ? 1.0 : 0.0);
            BAND_STAMP(2);          // gradient, H = rho J^T J + diag - M^T M
            if (!chol_g<G, NT>(H, rd, r) && bad == 0) bad = t + 1;
            BAND_STAMP(3);          // Cholesky
            // ---- right-hand side: y_t = L_tt^-1 (-g_t - L_{t,t-1} y_{t-1})
            double y[1] = {inT ? -g : 0.0};
            if (t > 0) {
                double a = 0.0;
    #pragma unroll
                for (int k = 0; k < NT; ++k) a = fma(Mt[k], Gr::rb(yprev[0], k), a);
                y[0] -= a;
            }
            // ---- keep the knot's factor rows (banded form), then turn the registers into the unit-triangular form
            if (t + 1 < T) load_state(t + 1);
            double *o = fac + (long long)t * NT * C::ROW + r;
            if (live && inT) {
    #pragma unroll
                for (int c = 0; c < NT; ++c) o[(c) * NT] = H[0][c];
                o[(NT) * NT] = rd[0];
            }
            unit_lower<G, NT>(H, rd, r);
            const double rdm = inT ? rd[0] : 0.0, nrho_rd = -rho * rdm;
            y[0] = trsv_unit<G, NT>(H, y[0]) * rdm;
            // ---- M_t = L_tt^-1 H_{t+1,t}^T: column j of it is the distributed vector -rho J[j][:]
            double M[NX];
    #pragma unroll
            for (int j = 0; j < NX; ++j) M[j] = trsv_unit<G, NT>(H, Jc[j]) * nrho_rd;
            if (live && inT) {
    #pragma unroll
                for (int j = 0; j < NX; ++j) o[(NT + 1 + j) * NT] = M[j];
                P.upd[b * (long long)T * NT + t * NT + r] = y[0];        // y parked in the output
            }
    #pragma unroll
            for (int j = 0; j < NX; ++j) { Mprev[j] = M[j]; mu_prev[j] = mu[j]; }
            yprev[0] = inT ? y[0] : 0.0;
            // M_t transposed through LDS for the next knot (read after that knot's model evaluation); a workgroup is one
            // wavefront: DS operations complete in order
    #pragma unroll
            for (int j = 0; j < NX; ++j) trow[j * TS] = M[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            BAND_STAMP(4);          // forward substitutions (y, M), factor stores, transposed copy
        }
    } else {
        // ---- a knot's inputs.  Small models (nt <= 8, where the registers allow it) load knot t + 1 while knot t
        // computes and write knot t's factor rows while knot t + 1 computes: a knot was one exposed memory round trip
        // in front of ~2.5 k instructions, and its stores were in the way of the next knot's loads (vmcnt is in order).
        const int rr = inT ? r : 0, rx = r < NX ? r : 0, iu = (inT && r >= NX) ? r - NX : 0;
        const double x0r = x0[rx], lam_first = lam[(T - 1) * NX + rx], uur = P.uu[iu], ulr = P.ul[iu];       // knot-independent
        // KB ("knot batching"): a group of G lanes has G / nt times the lanes one knot's nt forward-mode seeds need -- three
        // knots' worth at cartpole-1 on a 16-lane row, five at the pendulums -- and the model evaluation is the larger part
        // of a knot (1.1 k of 1.8 k instructions at cartpole-1).  Every KP knots the lanes evaluate the model of KP knots at
        // once (lane r: knot t + r / nt, seed r % nt) and leave the Jacobian columns and the residuals in LDS; the sweep
        // then takes each knot's column from there.
        constexpr int KP = !is_given<Map>::value ? G / NT : 1;
        constexpr bool KB = KP >= 2;
        constexpr int JBK = NX * (NT + 1);                     // per knot: NX rows of [J[j][0..nt-1], res_j]
        __shared__ double jbuf[KB ? (64 / G) * KP * JBK : 1];
        double *jb = jbuf + (KB ? (lane / G) * KP * JBK : 0);
        const int kk = KB ? r / NT : 0, kr = KB ? r - kk * NT : r;       // this lane's knot in the batch / its seed
        const bool kact = KB && r < KP * NT;
        double pzk[NT], pxnk[NX];                                        // the lane's model inputs of the NEXT batch
        auto load_batch = [&](int t0) {
            const int tk = t0 + kk, tc = (kact && tk < T) ? tk : 0, tn = (kact && tk < T - 1) ? tk + 1 : 0;
    #pragma unroll
            for (int j = 0; j < NT; ++j) pzk[j] = xu[tc * NT + j];
    #pragma unroll
            for (int j = 0; j < NX; ++j) pxnk[j] = xu[tn * NT + j];
        };
        double pz[NT], pxn1[NX], plam[NX], pfx[NX], pcol[NX], pzr, pqd, pq, plu, pll;
        auto load_knot = [&](int t) {
            const bool dynrow = t < T - 1;
            if constexpr (!KB) {
    #pragma unroll
                for (int j = 0; j < NT; ++j) pz[j] = xu[t * NT + j];
            }
    #pragma unroll
            for (int j = 0; j < NX; ++j) {
                if constexpr (!KB) pxn1[j] = dynrow ? xu[(t + 1) * NT + j] : 0.0;
                plam[j] = dynrow ? lam[t * NX + j] : 0.0;
            }
            if constexpr (is_given<Map>::value) {
                const int tc = dynrow ? t : 0;
                const double *fx = P.xnext + (b * (long long)(T - 1) + tc) * NX;
                const double *jx = P.Jx + (b * (long long)(T - 1) + tc) * NX * NX;
                const double *ju = P.Ju + (b * (long long)(T - 1) + tc) * NX * NU;
    #pragma unroll
                for (int j = 0; j < NX; ++j) {
                    pfx[j] = fx[j];
                    pcol[j] = r < NX ? jx[j * NX + r] : ju[j * NU + (inT ? r - NX : 0)];
                }
            }
            pzr = xu[t * NT + rr]; pqd = Qd[t * NT + rr]; pq = q[t * NT + rr];
            const int row = neq + t * 2 * NU + iu;
            plu = lam[row]; pll = lam[row + NU];
        };
        double sH[NT], sM[NX], srd = 0.0, sy = 0.0;              // knot t - 1's factor rows, stored during knot t
        auto store_knot = [&](int t, const double (&Hrow)[NT], double rdv, const double (&Mrow)[NX], double yv) {
            if (live && inT) {
                double *o = fac + (long long)t * NT * C::ROW + r;
    #pragma unroll
                for (int c = 0; c < NT; ++c) o[(c) * NT] = Hrow[c];
                o[(NT) * NT] = rdv;
    #pragma unroll
                for (int j = 0; j < NX; ++j) o[(NT + 1 + j) * NT] = Mrow[j];
                P.upd[b * (long long)T * NT + t * NT + r] = yv;        // y parked in the output
            }
        };
        load_knot(0);
        if constexpr (KB) load_batch(0);
    #pragma unroll
        for (int j = NX; j < G; ++j) trow[j * TS] = 0.0;

        for (int t = 0; t < T; ++t) {
            // ---- the knot and its successor's state
            double z[NT], xn1[NX], lamt[NX], fxv[NX], colv[NX];
            if constexpr (!KB) {
    #pragma unroll
                for (int j = 0; j < NT; ++j) z[j] = pz[j];
    #pragma unroll
                for (int j = 0; j < NX; ++j) xn1[j] = pxn1[j];
            }
    #pragma unroll
            for (int j = 0; j < NX; ++j) { lamt[j] = plam[j]; fxv[j] = pfx[j]; colv[j] = pcol[j]; }
            const double zr = pzr, qdr = pqd, qr = pq, lup = plu, llo = pll;
            const bool dynrow = t < T - 1;
            if (t + 1 < T) load_knot(t + 1);
            if constexpr (!LF) { if (t > 0) store_knot(t - 1, sH, srd, sM, sy); }
            // ---- column r of J_t = [df/dx df/du] by one forward-mode seed per lane
            double Jc[NX], mu[NX];
            if constexpr (is_given<Map>::value) {
    #pragma unroll
                for (int j = 0; j < NX; ++j) {
                    Jc[j] = (dynrow && inT) ? colv[j] : 0.0;
                    const double res = dynrow ? xn1[j] - fxv[j] : 0.0;
                    mu[j] = dynrow ? lamt[j] + rho * res : 0.0;
                }
            } else if constexpr (KB) {
                if (t % KP == 0) {            // the model of knots t .. t + KP - 1, one seed of one knot per lane
                    Dual<1> xs[NX], us[NU], out[NX];
    #pragma unroll
                    for (int j = 0; j < NX; ++j) { xs[j] = Dual<1>(pzk[j]); xs[j].d[0] = (kr == j) ? 1.0 : 0.0; }
    #pragma unroll
                    for (int j = 0; j < NU; ++j) { us[j] = Dual<1>(pzk[NX + j]); us[j].d[0] = (kr == NX + j) ? 1.0 : 0.0; }
                    Map::template step<Dual<1>>(xs, us, P.dt, out);
                    const bool dk = kact && t + kk < T - 1;           // the lane's knot has a dynamics row
                    if (kact) {
    #pragma unroll
                        for (int j = 0; j < NX; ++j) {
                            jb[kk * JBK + j * (NT + 1) + kr] = dk ? out[j].d[0] : 0.0;
                            if (kr == 0) jb[kk * JBK + j * (NT + 1) + NT] = dk ? pxnk[j] - out[j].v : 0.0;   // x_{t+1} - f(x_t, u_t)
                        }
                    }
                    if (t + KP < T) load_batch(t + KP);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                const int kt = t % KP;
    #pragma unroll
                for (int j = 0; j < NX; ++j) {
                    Jc[j] = inT ? jb[kt * JBK + j * (NT + 1) + (inT ? r : 0)] : 0.0;
                    mu[j] = dynrow ? lamt[j] + rho * jb[kt * JBK + j * (NT + 1) + NT] : 0.0;
                }
            } else {
                Dual<1> xs[NX], us[NU], out[NX];
    #pragma unroll
                for (int j = 0; j < NX; ++j) { xs[j] = Dual<1>(z[j]); xs[j].d[0] = (r == j) ? 1.0 : 0.0; }
    #pragma unroll
                for (int j = 0; j < NU; ++j) { us[j] = Dual<1>(z[NX + j]); us[j].d[0] = (r == NX + j) ? 1.0 : 0.0; }
                Map::template step<Dual<1>>(xs, us, P.dt, out);
    #pragma unroll
                for (int j = 0; j < NX; ++j) {
                    Jc[j] = (dynrow && inT) ? out[j].d[0] : 0.0;
                    const double res = dynrow ? xn1[j] - out[j].v : 0.0;                 // x_{t+1} - f(x_t, u_t)
                    mu[j] = dynrow ? lamt[j] + rho * res : 0.0;
                }
            }
            // ---- gradient element r of this knot, and the diagonal terms of H_tt
            double g = qdr * zr + qr, dg = qdr;
    #pragma unroll
            for (int j = 0; j < NX; ++j) g -= Jc[j] * mu[j];
            {
                // x rows: + I block of the previous dynamics rows / the x_0 rows; u rows: the box rows
                double prev = 0.0;
    #pragma unroll
                for (int j = 0; j < NX; ++j) prev = (r == j) ? mu_prev[j] : prev;
                const double first = lam_first + rho * (zr - x0r);
                const double rup = zr - uur, rlo = ulr - zr;
                if (r < NX) {
                    g += (t > 0) ? prev : first;
                    dg += rho;
                } else if (inT) {
                    g += (lup + rho * fmax(rup, 0.0)) - (llo + rho * fmax(rlo, 0.0));
                    dg += rho * ((rup > 0.0 ? 1.0 : 0.0) + (rlo > 0.0 ? 1.0 : 0.0));
                }
            }
            // ---- H_tt row r:  rho J^T J + diag - Gram(M_prev) on the x-x block
            double H[1][NT], rd[1];
    #pragma unroll
            for (int c = 0; c < NT; ++c) {
                double a = 0.0;
    #pragma unroll
                for (int j = 0; j < NX; ++j) a = fma(Jc[j], Gr::rb(Jc[j], c), a);
                H[0][c] = rho * a + ((r == c) ? dg : 0.0);
            }
            double Mt[G];              // row r of M_prev^T M_prev and M_prev^T y_prev from the transposed copy (as above)
            if (t > 0) {
    #pragma unroll
                for (int k = 0; k < G; ++k) Mt[k] = tcol[k];
    #pragma unroll
                for (int j = 0; j < NX; ++j) {
                    double a = 0.0;
    #pragma unroll
                    for (int k = 0; k < NT; ++k) a = fma(Mt[k], Gr::rb(Mprev[j], k), a);
                    H[0][j] -= a;
                }
            }
    #pragma unroll
            for (int c = 0; c < NT; ++c) H[0][c] = inT ? H[0][c] : ((r == c) ? 1.0 : 0.0);
            if (!chol_g<G, NT>(H, rd, r) && bad == 0) bad = t + 1;
            // ---- right-hand side: y_t = L_tt^-1 (-g_t - L_{t,t-1} y_{t-1})
            double y[1] = {inT ? -g : 0.0};
            if (t > 0) {
                double a = 0.0;
    #pragma unroll
                for (int k = 0; k < NT; ++k) a = fma(Mt[k], Gr::rb(yprev[0], k), a);
                y[0] -= a;
            }
            // ---- keep the knot's factor rows (banded form; written while the next knot computes), then the solves on the
            // unit-triangular form
            if constexpr (LF) {
                if (inT) {
    #pragma unroll
                    for (int c = 0; c < NT; ++c) lf[(t * LROW + c) * NL] = H[0][c];
                    lf[(t * LROW + NT) * NL] = rd[0];
                }
            } else {
    #pragma unroll
                for (int c = 0; c < NT; ++c) sH[c] = H[0][c];
            }
            unit_lower<G, NT>(H, rd, r);
            const double rdm = inT ? rd[0] : 0.0, nrho_rd = -rho * rdm;
            y[0] = trsv_unit<G, NT>(H, y[0]) * rdm;
            // ---- M_t = L_tt^-1 H_{t+1,t}^T: column j of it is the distributed vector -rho J[j][:]
            double M[NX];
    #pragma unroll
            for (int j = 0; j < NX; ++j) M[j] = trsv_unit<G, NT>(H, Jc[j]) * nrho_rd;
            if constexpr (LF) {
                if (inT) {
    #pragma unroll
                    for (int j = 0; j < NX; ++j) lf[(t * LROW + NT + 1 + j) * NL] = M[j];
                    lf[(t * LROW + C::ROW) * NL] = y[0];
                }
            } else {
    #pragma unroll
                for (int j = 0; j < NX; ++j) sM[j] = M[j];
                srd = rd[0]; sy = y[0];
            }
    #pragma unroll
            for (int j = 0; j < NX; ++j) { Mprev[j] = M[j]; mu_prev[j] = mu[j]; }
            yprev[0] = inT ? y[0] : 0.0;
    #pragma unroll
            for (int j = 0; j < NX; ++j) trow[j * TS] = M[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (!LF) store_knot(T - 1, sH, srd, sM, sy);
    }
    // ---- backward sweep: upd_t = L_tt^-T (y_t - M_t upd_{t+1}[:NX]); knot t - 1's rows are loaded while knot t is
    // solved (a knot here is a memory round trip in front of nt group sums)
    double xnext[1] = {0.0};
    double pL[NT], pM[NX], prd, pv;
    auto load_rows = [&](int t) {
        if constexpr (LF) {
#pragma unroll
            for (int c = 0; c < NT; ++c) pL[c] = lf[(t * LROW + c) * NL];
            prd = lf[(t * LROW + NT) * NL];
#pragma unroll
            for (int j = 0; j < NX; ++j) pM[j] = lf[(t * LROW + NT + 1 + j) * NL];
            pv = lf[(t * LROW + C::ROW) * NL];
        } else {
            const double *o = fac + (long long)t * NT * C::ROW + (inT ? r : 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) pL[c] = o[(c) * NT];
            prd = o[(NT) * NT];
#pragma unroll
            for (int j = 0; j < NX; ++j) pM[j] = o[(NT + 1 + j) * NT];
            pv = P.upd[b * (long long)T * NT + t * NT + (inT ? r : 0)];
        }
    };
    load_rows(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        double L[1][NT], rd[1], M[NX];
#pragma unroll
        for (int c = 0; c < NT; ++c) L[0][c] = inT ? pL[c] : 0.0;
        rd[0] = inT ? prd : 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) M[j] = inT ? pM[j] : 0.0;
        double v[1] = {inT ? pv : 0.0};
        if (t > 0) load_rows(t - 1);
        if (t < T - 1) {
#pragma unroll
            for (int j = 0; j < NX; ++j) v[0] = fma(-M[j], Gr::rb(xnext[0], j), v[0]);
        }
        v[0] = trsvT_bcast<G, NT, TS>(L, rd[0], v[0], trow, tcol, r);
        if (live && inT) P.upd[b * (long long)T * NT + t * NT + r] = v[0];
        xnext[0] = inT ? v[0] : 0.0;
    }
    if constexpr (LF) {
        if (P.keep && live && inT) {
            for (int t = 0; t < T; ++t) {
                double *o = fac + (long long)t * NT * C::ROW + r;
#pragma unroll
                for (int k = 0; k < C::ROW; ++k) o[k * NT] = lf[(t * LROW + k) * NL];
            }
        }
    }
    BAND_STAMP(5);                  // backward sweep
#ifdef DQP_BAND_STAMPS
    if (P.stamps && lane == 0) for (int k_ = 0; k_ < 8; ++k_) P.stamps[blockIdx.x * 8 + k_] = stamp_acc[k_];
#endif
    if (live && r == 0 && P.info) P.info[b] = bad;
}

// out = -(L L^T)^-1 rhs with the banded factor a forward launch left in `fac`
// (NewtonAL.backward, al_utils.py:477-480).  Both sweeps are chains of knots, each a memory round trip in front of a
// short substitution: the next knot's rows are in flight while one is solved, and the substitutions are the
// broadcast forms of the Newton kernel (unit_lower / trsv_unit forward, trsvT_bcast backward, M^T y from the LDS tile).
template <class Map, int G>
__global__ __launch_bounds__(64) void al_banded_solve_kernel(BandP P)
{
    using C = BandCfg<Map>;
    using Gr = Grp<G>;
    constexpr int NX = C::NX, NT = C::NT;
    static_assert(NT <= G, "a knot must fit its lane group");
    const int lane = threadIdx.x, r = lane & (G - 1);
    long long b = (long long)blockIdx.x * (64 / G) + lane / G;
    const bool live = b < P.B;
    if (!live) b = P.B - 1;
    const int T = P.T;
    const bool inT = r < NT;
    const double *fac = P.fac + b * (long long)T * NT * C::ROW;
    const double *rhs = P.rhs + b * (long long)T * NT;
    double *upd = P.upd + b * (long long)T * NT;
    constexpr int TS = G + 2, PS = G * TS + (G == 16 ? 16 : 0);
    __shared__ __attribute__((aligned(16))) double tile[(64 / G) * PS];
    double *trow = tile + (lane / G) * PS + r;
    const double *tcol = tile + (lane / G) * PS + r * TS;
#pragma unroll
    for (int j = 0; j < G; ++j) trow[j * TS] = 0.0;

    double pL[NT], pM[NX], prd, pv;
    auto load_rows = [&](int t, const double *vec, double sign) {
        const double *o = fac + (long long)t * NT * C::ROW + (inT ? r : 0);
#pragma unroll
        for (int c = 0; c < NT; ++c) pL[c] = o[(c) * NT];
        prd = o[(NT) * NT];
#pragma unroll
        for (int j = 0; j < NX; ++j) pM[j] = o[(NT + 1 + j) * NT];
        pv = sign * vec[t * NT + (inT ? r : 0)];
    };
    // ---- forward: y_t = L_tt^-1 (-rhs_t - M_{t-1}^T y_{t-1})
    double Mt[G], yprev = 0.0;
    load_rows(0, rhs, -1.0);
    for (int t = 0; t < T; ++t) {
        double L[1][NT], rd[1], M[NX];
#pragma unroll
        for (int c = 0; c < NT; ++c) L[0][c] = inT ? pL[c] : 0.0;
        rd[0] = inT ? prd : 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) M[j] = inT ? pM[j] : 0.0;
        double y = inT ? pv : 0.0;
        if (t + 1 < T) load_rows(t + 1, rhs, -1.0);
        if (t > 0) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < NT; ++k) a = fma(Mt[k], Gr::rb(yprev, k), a);
            y -= a;
        }
        unit_lower<G, NT>(L, rd, r);
        y = trsv_unit<G, NT>(L, y) * rd[0];
        if (live && inT) upd[t * NT + r] = y;
        yprev = inT ? y : 0.0;
        // column r of M_t for the next knot (lanes beyond nx read the zero columns)
#pragma unroll
        for (int j = 0; j < NX; ++j) trow[j * TS] = M[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < G; ++k) Mt[k] = tcol[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- backward: out_t = L_tt^-T (y_t - M_t out_{t+1}[:nx])
    double xnext = 0.0;
    load_rows(T - 1, upd, 1.0);
    for (int t = T - 1; t >= 0; --t) {
        double L[1][NT], M[NX];
#pragma unroll
        for (int c = 0; c < NT; ++c) L[0][c] = inT ? pL[c] : 0.0;
        const double rdv = inT ? prd : 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) M[j] = inT ? pM[j] : 0.0;
        double v = inT ? pv : 0.0;
        if (t > 0) load_rows(t - 1, upd, 1.0);
        if (t < T - 1) {
#pragma unroll
            for (int j = 0; j < NX; ++j) v = fma(-M[j], Gr::rb(xnext, j), v);
        }
        v = trsvT_bcast<G, NT, TS>(L, rdv, v, trow, tcol, r);
        if (live && inT) upd[t * NT + r] = v;
        xnext = inT ? v : 0.0;
    }
}

// Eight problems per wavefront where a knot fits a half row AND the batch is more than one wavefront per SIMD at
// four per wavefront (these kernels hold one wavefront per SIMD): below that the launch is one latency chain per
// wavefront either way and the narrower group only idles SIMDs (cartpole-1 at B = 4096: 2.0 ms both ways;
// cartpole-2 at B = 8192: 2.24 -> 1.97 ms per AL_mpc.MPC call).
template <class Map> constexpr bool half_row() { return Map::NX + Map::NU <= 8; }
// g_lane_group: 0 = by batch size, 8 / 16 = pinned (dqp_al_lane_group; initial value from DQP_AL_LANE_GROUP, read once)
static int g_lane_group = -1;
static int g_simds = 0;
inline bool narrow(int B)
{
    if (g_lane_group < 0) {
        const char *force = getenv("DQP_AL_LANE_GROUP");
        g_lane_group = (force && force[0] == '8') ? 8 : ((force && force[0] == '1') ? 16 : 0);
    }
    if (g_lane_group) return g_lane_group == 8;
    if (!g_simds) {         // four SIMDs per CU, one wavefront of these kernels per SIMD
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        g_simds = 4 * cus;
    }
    return (B + 3) / 4 > g_simds;
}
template <class Map> int run_newton(const BandP &P_, void *stream)
{
    BandP P = P_;
#ifdef DQP_BAND_STAMPS
    P.stamps = g_band_stamps;
#endif
    if constexpr (half_row<Map>()) {
        if (narrow(P.B)) {
            if constexpr (Map::NX + Map::NU >= 6) {
                // the one-wavefront-per-SIMD half-row models: the factor stays in LDS where the horizon allows (four
                // workgroups per CU next to their 5 KB tiles)
                using C = BandCfg<Map>;
                const size_t bytes = (size_t)P.T * (C::ROW + 1) * 8 * C::NT * sizeof(double);
                if (bytes <= 34 * 1024) {
                    DQP_LAUNCH((al_banded_newton_kernel<Map, 8, true>), dim3((P.B + 7) / 8), dim3(64), bytes, (hipStream_t)stream, P);
                    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
                }
            }
            DQP_LAUNCH((al_banded_newton_kernel<Map, 8>), dim3((P.B + 7) / 8), dim3(64), 0, (hipStream_t)stream, P);
            return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
        }
    }
    DQP_LAUNCH((al_banded_newton_kernel<Map, 16>), dim3((P.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}
template <class Map> int run_solve(const BandP &P, void *stream)
{
    if constexpr (half_row<Map>()) {
        if (narrow(P.B)) {
            DQP_LAUNCH((al_banded_solve_kernel<Map, 8>), dim3((P.B + 7) / 8), dim3(64), 0, (hipStream_t)stream, P);
            return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
        }
    }
    DQP_LAUNCH((al_banded_solve_kernel<Map, 16>), dim3((P.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

// (n_state, n_ctrl) pairs with a caller-linearised instantiation
#define DQP_BAND_SIZES                                                                                   \
    X(1, 1) X(2, 1) X(3, 1) X(4, 1) X(5, 1) X(6, 1) X(7, 1) X(8, 1) X(2, 2) X(3, 2) X(4, 2) X(5, 2) X(6, 2) X(8, 2) \
    X(10, 2) X(12, 2) X(3, 3) X(6, 3) X(9, 3) X(4, 4) X(6, 4) X(8, 4) X(10, 4) X(12, 4)

bool given_supported(int n, int m)
{
#define X(a, b) if (n == a && m == b) return true;
    DQP_BAND_SIZES
#undef X
    return false;
}

int knot_doubles(int id, int n_, int m_)          // nt rows x (L row, 1/diag, M row)
{
    int32_t n = n_, m = m_;
    if (id != 0 && dqp_dyn_sizes(id, &n, &m) != DQP_OK) return 0;
    if (id == 0 && !given_supported(n, m)) return 0;
    return (n + m) * ((n + m) + 1 + n);
}

}  // namespace

using namespace dqp::dyn;

extern "C" {

#ifdef DQP_BAND_STAMPS
// instrumented build only: device buffer of 8 x uint64 per workgroup of the Newton kernel
__attribute__((visibility("default"))) void dqp_debug_band_stamps(void *dev_ptr) { g_band_stamps = (unsigned long long *)dev_ptr; }
#endif

__attribute__((visibility("default"))) size_t dqp_al_banded_factor_bytes(const dqp_al_mpc_dims *d, int dyn_id)
{
    const int kd = d ? knot_doubles(dyn_id, d->n_state, d->n_ctrl) : 0;
    if (!d || d->nbatch <= 0 || d->T < 2 || kd == 0) return 0;
    return (size_t)d->nbatch * d->T * kd * sizeof(double);
}

}  // extern "C"

// dqp_al_banded_newton_step with the caller saying whether it needs this step's factor afterwards (the Newton loop of
// dqp_al.hip keeps the last step's only)
int dqp::al_banded_newton_step_keep(const dqp_al_mpc_dims *d, int dyn_id, double dt, const double *xu, const double *x0,
                                    const double *Qdiag, const double *q, const double *lam, const double *rho,
                                    const double *u_lower, const double *u_upper, double *update, void *factor,
                                    int32_t *info, void *stream, int keep)
{
    if (!d || d->nbatch < 0 || d->T < 2) return DQP_ERR_BAD_ARG;
    int32_t n = 0, m = 0;
    if (dqp_dyn_sizes(dyn_id, &n, &m) != DQP_OK || n != d->n_state || m != d->n_ctrl) return DQP_ERR_BAD_ARG;
    if (d->nbatch == 0) return DQP_OK;
    if (!xu || !x0 || !Qdiag || !q || !lam || !rho || !u_lower || !u_upper || !update || !factor) return DQP_ERR_BAD_ARG;
    BandP P = {xu, x0, Qdiag, q, lam, rho, u_lower, u_upper, nullptr, nullptr, nullptr, nullptr, update, (double *)factor,
               info, dt, d->nbatch, d->T, keep};
    switch (dyn_id) {
    case DQP_DYN_PENDULUM1L: return run_newton<Robot<Pendulum1l>>(P, stream);
    case DQP_DYN_CARTPOLE1L: return run_newton<Robot<Cartpole1l>>(P, stream);
    case DQP_DYN_CARTPOLE2L: return run_newton<Robot<Cartpole2l>>(P, stream);
    case DQP_DYN_PENDULUM_EULER: return run_newton<PendulumEuler>(P, stream);
    case DQP_DYN_REXQUADROTOR: return run_newton<RexQuadrotor>(P, stream);
    default: return run_newton<PendulumDx>(P, stream);
    }
}

extern "C" {

__attribute__((visibility("default"))) int
dqp_al_banded_newton_step(const dqp_al_mpc_dims *d, int dyn_id, double dt, const double *xu, const double *x0,
                          const double *Qdiag, const double *q, const double *lam, const double *rho,
                          const double *u_lower, const double *u_upper, double *update, void *factor,
                          int32_t *info, void *stream)
{
    return dqp::al_banded_newton_step_keep(d, dyn_id, dt, xu, x0, Qdiag, q, lam, rho, u_lower, u_upper, update, factor, info,
                                           stream, 1);
}

__attribute__((visibility("default"))) int
dqp_al_banded_solve(const dqp_al_mpc_dims *d, int dyn_id, const void *factor, const double *rhs, double *out,
                    void *stream)
{
    if (!d || d->nbatch < 0 || d->T < 2) return DQP_ERR_BAD_ARG;
    int32_t n = 0, m = 0;
    if (dyn_id != 0 && (dqp_dyn_sizes(dyn_id, &n, &m) != DQP_OK || n != d->n_state || m != d->n_ctrl)) return DQP_ERR_BAD_ARG;
    if (d->nbatch == 0) return DQP_OK;
    if (!factor || !rhs || !out) return DQP_ERR_BAD_ARG;
    BandP P = {};
    P.rhs = rhs; P.upd = out; P.fac = (double *)factor; P.B = d->nbatch; P.T = d->T;
    if (dyn_id == 0) {          // the factor of dqp_al_banded_newton_step_jac: layout by sizes
#define X(a, b) if (d->n_state == a && d->n_ctrl == b) return run_solve<Given<a, b>>(P, stream);
        DQP_BAND_SIZES
#undef X
        return DQP_ERR_TOO_LARGE;
    }
    switch (dyn_id) {
    case DQP_DYN_PENDULUM1L: return run_solve<Robot<Pendulum1l>>(P, stream);
    case DQP_DYN_CARTPOLE1L: return run_solve<Robot<Cartpole1l>>(P, stream);
    case DQP_DYN_CARTPOLE2L: return run_solve<Robot<Cartpole2l>>(P, stream);
    case DQP_DYN_PENDULUM_EULER: return run_solve<PendulumEuler>(P, stream);
    case DQP_DYN_REXQUADROTOR: return run_solve<RexQuadrotor>(P, stream);
    default: return run_solve<PendulumDx>(P, stream);
    }
}

/*
 * The same block-tridiagonal Newton step for a dynamics the CALLER linearised (include/dqp.h).
 */
__attribute__((visibility("default"))) int
dqp_al_banded_newton_step_jac(const dqp_al_mpc_dims *d, const double *xu, const double *x0, const double *Qdiag,
                              const double *q, const double *lam, const double *rho, const double *u_lower,
                              const double *u_upper, const double *x_next, const double *Jx, const double *Ju,
                              double *update, void *factor, int32_t *info, void *stream)
{
    if (!d || d->nbatch < 0 || d->T < 2 || d->n_state < 1 || d->n_ctrl < 1) return DQP_ERR_BAD_ARG;
    if (!given_supported(d->n_state, d->n_ctrl)) return DQP_ERR_TOO_LARGE;
    if (d->nbatch == 0) return DQP_OK;
    if (!xu || !x0 || !Qdiag || !q || !lam || !rho || !u_lower || !u_upper || !x_next || !Jx || !Ju || !update || !factor)
        return DQP_ERR_BAD_ARG;
    BandP P = {xu, x0, Qdiag, q, lam, rho, u_lower, u_upper, nullptr, x_next, Jx, Ju, update, (double *)factor, info, 0.0,
               d->nbatch, d->T, 1};
#define X(a, b) if (d->n_state == a && d->n_ctrl == b) return run_newton<Given<a, b>>(P, stream);
    DQP_BAND_SIZES
#undef X
    return DQP_ERR_TOO_LARGE;
}

__attribute__((visibility("default"))) int dqp_al_lane_group(int width)
{
    if (width != 0 && width != 8 && width != 16) return DQP_ERR_BAD_ARG;
    g_lane_group = width;
    return DQP_OK;
}

}  // extern "C"
