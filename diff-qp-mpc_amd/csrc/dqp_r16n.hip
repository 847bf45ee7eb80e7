// dqp_r16n.hip -- DPP-row kernels, null-space form (4 QPs per wavefront): the default forward and
// the backward that restarts from the forward's factorisation context.
//
// Same layout idea as dqp_r16.hip (one QP per 16-lane DPP row, matrices row-distributed in
// registers, v_mov_b64_dpp row_newbcast as the only cross-lane primitive) but the equality
// constraints are eliminated ONCE in setup, so the PDIPM iteration carries no equality rows:
//
//   hat coordinates   xh = Lq^T x (Q = Lq Lq^T),  Gh = G Lq^-T,  Ah = A Lq^-T
//   reverse LQ        Ah Qf = [0 | U]   (Householder reflectors H_k, Qf = H_{E-1} ... H_0,
//                                        U upper triangular E x E in the LAST E columns)
//   null-space split  xh = Qf [w ; xy],  xy = U^-1 b fixed,  Gh Qf = [Gz | W],  h' = h - W xy
//   reduced QP        min 1/2 |w|^2 + cp^T w   s.t.  Gz w + s = h',  s >= 0        (R = nz - neq)
//
// and the iteration is the one of dqp_r16.hip with no equality block: the nineq x nineq Schur
// complement T = Gz Gz^T + diag(s/z) is factored per iteration (the well-conditioned form: the
// reduced-Hessian alternative I + Gz^T D Gz was tried and loses ~1e-5 to cancellation), with
// three 30 x 15 mat-vecs instead of three 30 x 30 and five 15 x 30 ones, and the equality
// rows gone from the register file.
//
// The PDIPM iterates (x, s, z) of the reference (batch.py:46-208) are reproduced exactly in
// exact arithmetic: Newton's method is affine invariant and, with A x = b holding from the
// first iterate (batch.py:60-74), (dx, ds, dz) never depend on the equality multipliers y.
// The y iterate only enters the reference's stopping/selection residual through the range
// part of rx, which obeys rho_{k+1} = (1 - alpha_k) rho_k; we carry that scalar c_k and
// reconstruct  ||rx|| = ||(Lq Z) rw + c_k (Lq Y) rho_0||  and  y = U^-T (c rho_0 - xy - py - W^T z).
//
// LDS per QP: packed Lq, then a region that holds the reflector tails during setup and the
// epilogue and the packed Gz Gz^T during the iteration (the tails wait in the caller's
// workspace, dqp_workspace_bytes), then the parked vectors.
//
// The same workspace then carries the factorisation context (packed Lq, the tails, [Gz | W], U)
// to backward_kernel (-DDQP_R16_BWD objects), which therefore has no setup at all -- the
// counterpart of the reference keeping Q_LU / S_LU / R on ctx (qp.py:93-95).
//
// Reference functions covered: a2-a8 of SURVEY.md §8 (as dqp_pdipm.hip / dqp_r16.hip).

#include "dqp_r16_prims.h"

namespace dqp {
namespace r16n {

using namespace dqp::r16;

template <int N_, int M_, int E_> struct Cfg {
    static constexpr int N = N_, M = M_, E = E_, R = N_ - E_;
    static constexpr int SN = slots(N_), SM = slots(M_), SR = slots(N_ - E_);
    static constexpr int SE = E_ > 0 ? slots(E_) : 1, EC = E_ > 0 ? E_ : 1;
    static constexpr bool PARK = slots(N_) >= 3;        // W, U leave the registers between setup and epilogue
#ifndef DQP_R16N_PARK_WU
#define DQP_R16N_PARK_WU 1
#endif
    // the same for the two-slot sizes, without the split passes: W and U are dead between setup and epilogue and are
    // in the workspace anyway (the backward context); 1 drops them from the register file during the iteration
    static constexpr bool PARKWU = PARK || DQP_R16N_PARK_WU;
#ifndef DQP_R16N_PIN_ALL
#define DQP_R16N_PIN_ALL 1
#endif
    static constexpr bool PIN = PARK || DQP_R16N_PIN_ALL;   // ordered setup (pin / exec-masked tail stores)
#ifndef DQP_R16N_SPLIT_ALL
#define DQP_R16N_SPLIT_ALL 0
#endif
    static constexpr bool SPLIT = PARK || DQP_R16N_SPLIT_ALL;   // A pass (B, C, D on Ah) before the G pass
#ifndef DQP_R16N_EARLY_CTX
#define DQP_R16N_EARLY_CTX 1
#endif
    // the factorisation context goes to the workspace piece by piece as each piece becomes final, so
    // that its 16 KB per QP drain under the following phases instead of in one burst at the end
    static constexpr bool EARLY = DQP_R16N_EARLY_CTX;
    // LDS per QP (doubles)
    static constexpr int tailsz = E_ * (N_ - E_) + E_ * (E_ - 1) / 2;     // sum_k c_k, c_k = R + k
    static constexpr int oLq = 0;                       // packed lower triangle of Lq
    static constexpr int oTl = tri(N_);                 // packed reflector tails u'_k[0..c_k) ...
    static constexpr int oR = oTl;                      // ... / packed Gz Gz^T while iterating
    static constexpr int shared = tailsz > tri(M_) ? tailsz : tri(M_);
    static constexpr int oCp = oTl + shared;            // cp (R)
    static constexpr int oHp = oCp + (N_ - E_);         // h' (M)
    static constexpr int oQv = oHp + M_;                // qv = Lq Y (W^T 1)  (N)
    static constexpr int oBw = oQv + N_;                // best w (R), s (M), z (M), c (1)
    static constexpr int oBs = oBw + (N_ - E_), oBz = oBs + M_, oBc = oBz + M_;
    static constexpr int oDummy = oBc + 2;              // 16 write-only sink slots
    static constexpr int ldsQP = oDummy + 16;
    static constexpr int ldsQPpad = (ldsQP + 1) & ~1;
#ifndef DQP_R16N_STAGED
#define DQP_R16N_STAGED 0
#endif
    // Q, G, A through LDS with coalesced 16-byte-per-lane loads (a slot of 16 rows must fit the LDS behind
    // the packed Lq).  Measured at the metric size, B = 4096: 226.3 us per launch against 217.0 us with each lane
    // streaming its own row (load_rows) -- the copy, the two LDS hops and their waits cost more than the 4x
    // fewer cache-line lookups save -- so it stays off; -DDQP_R16N_STAGED=1 builds it.
    static constexpr bool STAGED = DQP_R16N_STAGED && 16 * N_ <= ldsQP - oTl;
    __host__ __device__ static constexpr int toff(int k) { return k * (N_ - E_) + k * (k - 1) / 2; }
    // caller workspace per QP (doubles): the reflector tails (parked during the iteration), then
    // the factorisation context the backward pass restarts from -- what the reference keeps on
    // ctx (Q_LU, S_LU, R: qp.py:93-95): packed Lq, [Gz | W] and U column-major, tau, 1/diag(U),
    // 1/diag(Lq)
    static constexpr int wTl = 0, wLq = tailsz, wG = wLq + tri(N_), wU = wG + M_ * N_;
    static constexpr int wTau = wU + E_ * E_, wRdu = wTau + E_, wRdq = wRdu + E_;
    // what the batch-rule finish pass needs on top of the backward context: xy, py, w1 (E each), delta
    static constexpr int wXy = wRdq + N_, wPy = wXy + E_, wW1 = wPy + E_, wDl = wW1 + E_;
    static constexpr int wsQP = wDl + 1;
    static constexpr int snapDim = (N_ - E_) + 2 * M_ + 2;      // w, s, z, c of an iterate (+ pad)
};

// Phase boundary for the compiler: LDS contents are to be re-read after this point.  Without it LLVM
// keeps every Lq entry a phase has read from LDS alive (in AGPRs, then scratch) to reuse it in a later
// phase that reads the same address -- for N = 40 that is the whole 820-entry triangle, i.e. the 13 KB
// of scratch per lane the 3-slot kernels used to have.
#define DQP_PHASE_FENCE() asm volatile("" ::: "memory")
// x passes through a side-effecting no-op: everything that produces it is scheduled before, every
// memory read that follows after.  (__builtin_amdgcn_sched_barrier alone orders memory operations;
// an independent FMA chain -- e.g. the second register slot of a triangular solve -- still floats
// past it and the LDS operands it needs later end up in scratch.)
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x) : : "memory"); }
// the same LDS address as an unrelated pointer, as far as the optimiser can tell
__device__ __forceinline__ const double *relabel(const double *p)
{
    asm volatile("" : "+v"(p));
    return p;
}

template <class C> struct State {
    double Gh[C::SM][C::N];     // Gh Qf = [Gz | W], row-distributed by constraint
    double LqZ[C::SN][C::R];    // (Lq Qf)[:, :R]
    double Ah[C::SE][C::N];     // U = Ah[:, R:]  (columns < c_k of row k are dead)
    double tau[C::SE];          // tau'_k, E-space distributed
    double rdu1[C::SE];         // 1 / U[k][k]
    double xy[C::SE];           // U^-1 b
    double py[C::SE];           // (Qf^T ph)[R:]  in E-space
    double w1[C::SE];           // W^T 1
    double rdq[C::SN];
    double rdiag[C::SM];        // diag(Gz Gz^T) of the lane's rows (factor_T rewrites the LDS copy)
    int status;
};

// element R+e of an N-space distributed vector -> element e of an E-space distributed vector
template <class C>
__device__ __forceinline__ void shift_down(const double (&vn)[C::SN], double (&ve)[C::SE], int r)
{
#pragma unroll
    for (int s = 0; s < C::SE; ++s) ve[s] = 0.0;
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const double t = BC(vn, C::R + e);
        if (r == (e & 15)) ve[e >> 4] = t;
    }
}
// E-space element e -> N-space position R+e (positions < R untouched)
template <class C>
__device__ __forceinline__ void shift_up(const double (&ve)[C::SE], double (&vn)[C::SN], int r)
{
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const double t = BC(ve, e);
        if (r == ((C::R + e) & 15)) vn[(C::R + e) >> 4] = t;
    }
}

// v <- H_k v for one reflector (u'_k read from LDS as a distributed vector)
template <class C>
__device__ __forceinline__ void reflect1(const double *lds, const double (&tau)[C::SE],
                                         double (&v)[C::SN], int k, int r)
{
    const int ck = C::R + k;
    const double *tl = lds + C::oTl + C::toff(k);
    double ud[C::SN], dot = 0.0;
#pragma unroll
    for (int s = 0; s < C::SN; ++s) {
        if (16 * s > ck) { ud[s] = 0.0; continue; }
        const int c = r + 16 * s;
        const double t = tl[c < ck ? c : 0];
        ud[s] = c < ck ? t : (c == ck ? 1.0 : 0.0);
        dot = fma(ud[s], v[s], dot);
    }
    const double tw = BC(tau, k) * row_sum(dot);
#pragma unroll
    for (int s = 0; s < C::SN; ++s)
        if (16 * s <= ck) v[s] = fma(-tw, ud[s], v[s]);
}
// v <- Qf^T v = H_0 ... H_{E-1} v   (H_{E-1} first)
template <class C>
__device__ __forceinline__ void apply_QfT(const double *lds, const double (&tau)[C::SE],
                                          double (&v)[C::SN], int r)
{
#pragma unroll
    for (int k = C::E - 1; k >= 0; --k) reflect1<C>(lds, tau, v, k, r);
}
// v <- Qf v = H_{E-1} ... H_0 v   (H_0 first)
template <class C>
__device__ __forceinline__ void apply_Qf(const double *lds, const double (&tau)[C::SE],
                                         double (&v)[C::SN], int r)
{
#pragma unroll
    for (int k = 0; k < C::E; ++k) reflect1<C>(lds, tau, v, k, r);
}

// t <- U^-T t  (U upper triangular E x E in Ah[:, R:], rows distributed in E-space)
template <class C>
__device__ __forceinline__ void solve_UT(const State<C> &st, double (&t)[C::SE], int r)
{
#pragma unroll
    for (int j = 0; j < C::E; ++j) {
        const int sj = j >> 4, lj = j & 15;
        double part = 0.0;
#pragma unroll
        for (int s = 0; s < C::SE; ++s) {
            if (16 * s >= j) continue;
            const int k = r + 16 * s;
            part = fma(k < j ? st.Ah[s][C::R + j] : 0.0, t[s], part);
        }
        const double tot = row_sum(part);
        if (r == lj) t[sj] = (t[sj] - tot) * st.rdu1[sj];
    }
}

// y = Gz v  (v in R-space) -> M-space
template <class C>
__device__ __forceinline__ void mul_Gz(const State<C> &st, const double (&v)[C::SR], double (&y)[C::SM])
{
#pragma unroll
    for (int s = 0; s < C::SM; ++s) y[s] = 0.0;
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const double vb = BC(v, t);
#pragma unroll
        for (int s = 0; s < C::SM; ++s) y[s] = fma(st.Gh[s][t], vb, y[s]);
    }
}
// y = Gz^T v (v in M-space) -> R-space   (transposed product: partials + reduce-scatter)
template <class C>
__device__ __forceinline__ void mul_GzT(const State<C> &st, const double (&v)[C::SM], double (&y)[C::SR], int r)
{
#pragma unroll
    for (int g = 0; g < C::SR; ++g) {
        double p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = 16 * g + k;
            if (c < C::R) {
                double a = st.Gh[0][c < C::R ? c : 0] * v[0];
#pragma unroll
                for (int s = 1; s < C::SM; ++s) a = fma(st.Gh[s][c < C::R ? c : 0], v[s], a);
                p[k] = a;
            } else {
                p[k] = 0.0;
            }
        }
        y[g] = reduce_scatter16(p, r);
    }
}
// y = W^T v (v in M-space) -> E-space   (transposed product: partials + reduce-scatter)
template <class C>
__device__ __forceinline__ void mul_WT(const State<C> &st, const double (&v)[C::SM], double (&y)[C::SE], int r)
{
#pragma unroll
    for (int g = 0; g < C::SE; ++g) {
        double p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = 16 * g + k;
            if (e < C::E) {
                double a = st.Gh[0][C::R + (e < C::E ? e : 0)] * v[0];
#pragma unroll
                for (int s = 1; s < C::SM; ++s) a = fma(st.Gh[s][C::R + (e < C::E ? e : 0)], v[s], a);
                p[k] = a;
            } else {
                p[k] = 0.0;
            }
        }
        y[g] = reduce_scatter16(p, r);
    }
}

// ---- MPC-structured loaders (qp_wrapper.py:638-679 evaluated in registers): row i of the dense
// Q / G / A built from the time-major (C, F) of problem qp.  Column c is a compile-time index after
// unrolling; its knot tc = c / nt and offset jc = c % nt are carried as uniform counters.
template <int S, int NC>
__device__ __forceinline__ void mpc_rows_Q(const KParams &P, long long qp, double (&dst)[S][NC], int r)
{
    const int nt = P.mn + P.mm;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = r + 16 * s, ic = i < NC ? i : NC - 1;
        const int t = ic / nt, j = ic - t * nt;
        const double *row = P.mC + (((long long)t * P.B + qp) * nt + j) * nt;
        int tc = 0, jc = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double v = row[jc];
            dst[s][c] = (t == tc && i < NC) ? v : 0.0;
            if (++jc == nt) { jc = 0; ++tc; }
        }
    }
}
template <int S, int NC>
__device__ __forceinline__ void mpc_rows_G(const KParams &P, int nrows, double (&dst)[S][NC], int r)
{
    const int n = P.mn, m = P.mm, nt = n + m, half = P.mT * m;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = r + 16 * s;
        const int k = i < half ? i : i - half;
        const int t = k / m, col = t * nt + n + (k - t * m);
        const double sg = i < half ? 1.0 : -1.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) dst[s][c] = (c == col && i < nrows) ? sg : 0.0;
    }
}
template <int S, int NC>
__device__ __forceinline__ void mpc_rows_A(const KParams &P, long long qp, int nrows, double (&dst)[S][NC], int r)
{
    const int n = P.mn, nt = n + P.mm, T = P.mT;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int e = r + 16 * s, ec = e < nrows ? e : nrows - 1;
        const int t = ec / n, j = ec - t * n;
        const bool dynrow = t < T - 1;
        const double *row = P.mF + (((long long)(dynrow ? t : 0) * P.B + qp) * n + j) * nt;
        int tc = 0, jc = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double f = row[jc];
            double v;
            if (dynrow) v = (tc == t) ? f : ((tc == t + 1 && jc == j) ? -1.0 : 0.0);
            else v = (tc == 0 && jc == j) ? 1.0 : 0.0;
            dst[s][c] = e < nrows ? v : 0.0;
            if (++jc == nt) { jc = 0; ++tc; }
        }
    }
}

// [Gz | W], U, tau, 1/diag(U) -> the caller's workspace (the factorisation context backward restarts
// from).  Column-major: the 16 lanes of a QP write 16 consecutive doubles (row-major 8-byte stores
// 240 B apart cost 5x write amplification in HBM).
template <class C>
__device__ __forceinline__ void park_GU(double *ws, const State<C> &st, int r)
{
    constexpr int N = C::N, M = C::M, E = C::E, R = C::R;
#pragma unroll
    for (int s = 0; s < C::SM; ++s) {
        const int i = r + 16 * s;
        if (i < M) {
#pragma unroll
            for (int c = 0; c < N; ++c) ws[C::wG + c * M + i] = st.Gh[s][c];
        }
    }
#pragma unroll
    for (int s = 0; s < C::SE; ++s) {
        const int k = r + 16 * s;
        if (k < E) {
#pragma unroll
            for (int e = 0; e < E; ++e) ws[C::wU + e * E + k] = st.Ah[s][R + e];
            ws[C::wTau + k] = st.tau[s];
            ws[C::wRdu + k] = st.rdu1[s];
        }
    }
}
// W = Gh[:, R:] and U = Ah[:, R:] back from the workspace (3-slot sizes drop them from the register
// file between the setup and the epilogue)
template <class C>
__device__ __forceinline__ void unpark_WU(const double *ws, State<C> &st, int r)
{
    constexpr int M = C::M, E = C::E, R = C::R;
#pragma unroll
    for (int s = 0; s < C::SM; ++s) {
        const int i = r + 16 * s, ic = i < M ? i : M - 1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double v = ws[C::wG + (R + e) * M + ic];
            st.Gh[s][R + e] = i < M ? v : 0.0;
        }
    }
#pragma unroll
    for (int s = 0; s < C::SE; ++s) {
        const int k = r + 16 * s, kc = k < E ? k : E - 1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double v = ws[C::wU + e * E + kc];
            st.Ah[s][R + e] = k < E ? v : 0.0;
        }
    }
}

#ifdef DQP_SETUP_STOP
// keeps alive exactly what the setup has produced up to phase boundary K (registers; the LDS and
// workspace writes are side effects already)
template <int K, class C> __device__ __forceinline__ double stop_sum(const State<C> &st)
{
    double a = 0.0;
#pragma unroll
    for (int s = 0; s < C::SN; ++s) a += st.rdq[s];
    if (K >= 2 || K == 0) {
#pragma unroll
        for (int s = 0; s < C::SM; ++s)
#pragma unroll
            for (int c = 0; c < C::N; ++c) a += st.Gh[s][c];
#pragma unroll
        for (int s = 0; s < C::SE; ++s)
#pragma unroll
            for (int c = 0; c < C::N; ++c) a += st.Ah[s][c];
    }
    if (K >= 3 || K == 0) {
#pragma unroll
        for (int s = 0; s < C::SE; ++s) a += st.tau[s] + st.rdu1[s] + st.xy[s];
    }
    if (K >= 4 || K == 0) {
#pragma unroll
        for (int s = 0; s < C::SE; ++s) a += st.py[s] + st.w1[s];
    }
    if (K >= 5 || K == 0) {
#pragma unroll
        for (int s = 0; s < C::SN; ++s)
#pragma unroll
            for (int c = 0; c < C::R; ++c) a += st.LqZ[s][c];
    }
    if (K >= 15) {
#pragma unroll
        for (int s = 0; s < C::SM; ++s) a += st.rdiag[s];
    }
    return a;
}
#endif

// ------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void setup(const KParams &P, long long qp, int r, double *lds, State<C> &st)
{
    constexpr int N = C::N, M = C::M, E = C::E, R = C::R;
    constexpr int SN = C::SN, SM = C::SM, SE = C::SE, SR = C::SR;
    double *dummy = lds + C::oDummy + r;
    st.status = DQP_STATUS_OK;
    // the three right-hand-side vectors are fetched now and consumed in phases D-F, so their HBM
    // latency hides under the factorisations instead of serialising three more round trips
    double p0[SN], h0[SM], b0[SE];
    const bool mpc = P.mC != nullptr;
    if (mpc) {      // p = c, h = [u_upper ; -u_lower], b = [-f ; x0]   (qp_wrapper.py:638-679)
        const int n = P.mn, m = P.mm, nt = n + m, T = P.mT;
#pragma unroll
        for (int s = 0; s < SN; ++s) {
            const int i = r + 16 * s, ic = i < N ? i : N - 1, t = ic / nt;
            const double v = P.mc[((long long)t * P.B + qp) * nt + (ic - t * nt)];
            p0[s] = i < N ? v : 0.0;
        }
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            const int i = r + 16 * s, k = (i < T * m ? i : i - T * m), ju = (k < T * m ? k : 0) % m;
            const double v = i < T * m ? P.muu[ju] : -P.mul[ju];
            h0[s] = i < M ? v : 0.0;
        }
#pragma unroll
        for (int s = 0; s < SE; ++s) {
            const int e = r + 16 * s, ec = e < E ? e : (E > 0 ? E - 1 : 0), t = ec / n, j = ec - t * n;
            const double v = t < T - 1 ? -P.mf[((long long)t * P.B + qp) * n + j] : P.mx0[qp * n + j];
            b0[s] = (E > 0 && e < E) ? v : 0.0;
        }
    } else {
#pragma unroll
        for (int s = 0; s < SN; ++s) p0[s] = (r + 16 * s < N) ? P.p[qp * P.sp + r + 16 * s] : 0.0;
#pragma unroll
        for (int s = 0; s < SM; ++s) h0[s] = (r + 16 * s < M) ? P.h[qp * P.sh + r + 16 * s] : 0.0;
#pragma unroll
        for (int s = 0; s < SE; ++s) b0[s] = (E > 0 && r + 16 * s < E) ? P.b[qp * P.sb + r + 16 * s] : 0.0;
    }
    {   // A: Q -> Lq -> packed LDS
        double Lq[SN][N];
        if (mpc) mpc_rows_Q<SN, N>(P, qp, Lq, r);
        else if (C::STAGED) load_rows_staged<SN, N, (N * N) % 2 == 0>(P.Q + qp * P.sQ, N, Lq, r, lds + C::oTl);
        else load_rows<SN, N>(P.Q + qp * P.sQ, N, Lq, r);
        if (!chol_rows<SN, N>(Lq, st.rdq, r)) st.status = DQP_STATUS_Q_NOT_PD;
        tri_store<SN, N>(lds + C::oLq, Lq, r, dummy);
    }
    __syncthreads();
    if (C::EARLY) {
        double *ws = P.workspace + qp * (long long)C::wsQP;
        for (int e = r; e < tri(N); e += 16) ws[C::wLq + e] = lds[C::oLq + e];
#pragma unroll
        for (int s = 0; s < SN; ++s)
            if (r + 16 * s < N) ws[C::wRdq + r + 16 * s] = st.rdq[s];
    }
    __builtin_amdgcn_sched_barrier(0);
    DQP_PHASE_FENCE();
    STAMP(P, 1);

    // B: rows of G, A times Lq^-T (Lq[j][k] read row-uniformly from LDS).  Sizes with three register
    // slots per N-space row (C::PARK) cannot hold Gh, Ah and a reflector at once: they do A first
    // (B, C, D on Ah alone), then G (B, then the reflectors re-read from LDS) -- see below.
    if (mpc) {
        if (!C::SPLIT) mpc_rows_G<SM, N>(P, M, st.Gh, r);
        if (E > 0) mpc_rows_A<SE, N>(P, qp, E, st.Ah, r);
    } else {
        if (C::STAGED) {
            if (!C::SPLIT) load_rows_staged<SM, N, (M * N) % 2 == 0>(P.G + qp * P.sG, M, st.Gh, r, lds + C::oTl);
            if (E > 0) load_rows_staged<SE, N, (C::EC * N) % 2 == 0>(P.A + qp * P.sA, E, st.Ah, r, lds + C::oTl);
        } else {
            if (!C::SPLIT) load_rows<SM, N>(P.G + qp * P.sG, M, st.Gh, r);
            if (E > 0) load_rows<SE, N>(P.A + qp * P.sA, E, st.Ah, r);
        }
    }
    {
        const double *Lp = lds + C::oLq;
#pragma unroll
        for (int j = 0; j < N; ++j) {
#pragma unroll
            for (int k = 0; k < j; ++k) {
                const double ljk = Lp[tri(j) + k];
                if (!C::SPLIT) {
#pragma unroll
                    for (int s = 0; s < SM; ++s) st.Gh[s][j] = fma(-st.Gh[s][k], ljk, st.Gh[s][j]);
                }
                if (E > 0) {
#pragma unroll
                    for (int s = 0; s < SE; ++s) st.Ah[s][j] = fma(-st.Ah[s][k], ljk, st.Ah[s][j]);
                }
            }
            const double rj = BC(st.rdq, j);
            if (!C::SPLIT) {
#pragma unroll
                for (int s = 0; s < SM; ++s) st.Gh[s][j] *= rj;
            }
            if (E > 0) {
#pragma unroll
                for (int s = 0; s < SE; ++s) st.Ah[s][j] *= rj;
            }
            // keep each column's LDS reads next to their FMAs (otherwise the scheduler may drift
            // the Ah chain away from the Gh chain and spill the Lq values in between)
            if (C::PIN) {
#pragma unroll
                for (int s = 0; s < SE; ++s) pin(st.Ah[s][j]);
                if (!C::SPLIT) {
#pragma unroll
                    for (int s = 0; s < SM; ++s) pin(st.Gh[s][j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    DQP_PHASE_FENCE();
    STAMP(P, 2);

#pragma unroll
    for (int s = 0; s < SE; ++s) { st.tau[s] = 0.0; st.rdu1[s] = 0.0; st.xy[s] = 0.0; st.py[s] = 0.0; st.w1[s] = 0.0; }

    if (E > 0) {
        // C: reverse Householder LQ of Ah; reflectors applied to the rows of Ah (above k) and Gh
        double nmax = 0.0;
#pragma unroll
        for (int k = E - 1; k >= 0; --k) {
            const int sk = k >> 4, lk = k & 15, ck = R + k;
#ifdef DQP_STAMPS_C      /* experiment: one stamp per reflector */
            if (k < 15) { STAMPC(P, k); }
#endif
            double u[N];
            double nrm2 = 0.0;
#pragma unroll
            for (int c = 0; c <= ck; ++c) {
                u[c] = rb(st.Ah[sk][c], lk);
                nrm2 = fma(u[c], u[c], nrm2);
            }
            // |U[k][k]|^2 = nrm2: a row that (numerically) lies in the span of the rows below it
            if (!(nrm2 > 1e-26 * nmax)) { if (st.status == DQP_STATUS_OK) st.status = DQP_STATUS_A_RANK_DEF; nrm2 = 1.0; }
            nmax = fmax(nmax, nrm2);
            const double vk = u[ck];
            const double alpha = -copysign(sqrt(nrm2), vk);
            const double uck = vk - alpha;
            const double iu = frcp(uck);
            const double tauk = uck * uck * frcp(nrm2 - alpha * vk);
            double *tl = lds + C::oTl + C::toff(k);
            if (C::PIN) {
                // one exec-masked region with immediate LDS offsets (the select-an-address form below
                // costs an address VGPR per element, and LLVM then keeps all of them for phase G)
#pragma unroll
                for (int c = 0; c < ck; ++c) u[c] *= iu;
                if (r == 0) {
#pragma unroll
                    for (int c = 0; c < ck; ++c) tl[c] = u[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < ck; ++c) {
                    u[c] *= iu;
                    double *dst = (r == 0) ? tl + c : dummy;
                    *dst = u[c];
                }
            }
            if (r == lk) { st.tau[sk] = tauk; st.rdu1[sk] = frcp(alpha); }
            // rows of Ah above k
#pragma unroll
            for (int s = 0; s < SE; ++s) {
                if (16 * s > k) continue;
                double w = st.Ah[s][ck];
#pragma unroll
                for (int c = 0; c < ck; ++c) w = fma(st.Ah[s][c], u[c], w);
                double tw = tauk * w;
                if (16 * s + 15 >= k) tw = (r + 16 * s < k) ? tw : 0.0;
#pragma unroll
                for (int c = 0; c < ck; ++c) st.Ah[s][c] = fma(-tw, u[c], st.Ah[s][c]);
                st.Ah[s][ck] -= tw;
            }
            if (r == lk) st.Ah[sk][ck] = alpha;
            // rows of Gh
            if (!C::SPLIT) {
#pragma unroll
                for (int s = 0; s < SM; ++s) {
                    double w = st.Gh[s][ck];
#pragma unroll
                    for (int c = 0; c < ck; ++c) w = fma(st.Gh[s][c], u[c], w);
                    const double tw = tauk * w;
#pragma unroll
                    for (int c = 0; c < ck; ++c) st.Gh[s][c] = fma(-tw, u[c], st.Gh[s][c]);
                    st.Gh[s][ck] -= tw;
                }
            }
            if (C::PIN) {           // next reflector's LDS traffic stays behind this one's arithmetic
#pragma unroll
                for (int s = 0; s < SE; ++s) pin(st.Ah[s][ck]);
                if (!C::SPLIT) {
#pragma unroll
                    for (int s = 0; s < SM; ++s) pin(st.Gh[s][ck]);
                }
            }
        }
#ifdef DQP_STAMPS_C
        STAMPC(P, 15);
#endif
        __syncthreads();          // tails are read back as distributed vectors
        __builtin_amdgcn_sched_barrier(0);
        if (C::EARLY) {
            double *ws = P.workspace + qp * (long long)C::wsQP;
            for (int e = r; e < C::tailsz; e += 16) ws[C::wTl + e] = lds[C::oTl + e];
        }
        DQP_PHASE_FENCE();

        // D: xy = U^-1 b
        {
            double b[SE];
#pragma unroll
            for (int s = 0; s < SE; ++s) b[s] = b0[s];
#pragma unroll
            for (int j = E - 1; j >= 0; --j) {
                const int sj = j >> 4, lj = j & 15;
                const double xj = rb(b[sj] * st.rdu1[sj], lj);
#pragma unroll
                for (int s = 0; s < SE; ++s) {
                    if (16 * s > j) continue;
                    const int k = r + 16 * s;
                    b[s] = (k == j) ? xj : fma(k < j ? -st.Ah[s][R + j] : 0.0, xj, b[s]);
                }
            }
#pragma unroll
            for (int s = 0; s < SE; ++s) st.xy[s] = b[s];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    DQP_PHASE_FENCE();
    if (C::SPLIT) {
        // B, C for G on its own: Gh = G Lq^-T, then the reflectors (tails in LDS, read row-uniformly)
        if (mpc) mpc_rows_G<SM, N>(P, M, st.Gh, r);
        else if (C::STAGED) load_rows_staged<SM, N, (M * N) % 2 == 0>(P.G + qp * P.sG, M, st.Gh, r, lds + C::oTl);
        else load_rows<SM, N>(P.G + qp * P.sG, M, st.Gh, r);
        const double *Lp = relabel(lds + C::oLq);
#pragma unroll
        for (int j = 0; j < N; ++j) {
#pragma unroll
            for (int k = 0; k < j; ++k) {
                const double ljk = Lp[tri(j) + k];
#pragma unroll
                for (int s = 0; s < SM; ++s) st.Gh[s][j] = fma(-st.Gh[s][k], ljk, st.Gh[s][j]);
            }
            const double rj = BC(st.rdq, j);
#pragma unroll
            for (int s = 0; s < SM; ++s) { st.Gh[s][j] *= rj; pin(st.Gh[s][j]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (E > 0) {
#pragma unroll
            for (int k = E - 1; k >= 0; --k) {
                const int ck = R + k;
                const double *tl = lds + C::oTl + C::toff(k);
                const double tauk = BC(st.tau, k);
                double w[SM];
#pragma unroll
                for (int s = 0; s < SM; ++s) w[s] = st.Gh[s][ck];
#pragma unroll
                for (int c = 0; c < ck; ++c) {
                    const double uc = tl[c];
#pragma unroll
                    for (int s = 0; s < SM; ++s) w[s] = fma(st.Gh[s][c], uc, w[s]);
                }
#pragma unroll
                for (int s = 0; s < SM; ++s) { w[s] *= tauk; st.Gh[s][ck] -= w[s]; }
#pragma unroll
                for (int c = 0; c < ck; ++c) {
                    const double uc = tl[c];
#pragma unroll
                    for (int s = 0; s < SM; ++s) st.Gh[s][c] = fma(-w[s], uc, st.Gh[s][c]);
                }
#pragma unroll
                for (int s = 0; s < SM; ++s) { pin(st.Gh[s][0]); pin(st.Gh[s][ck]); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    DQP_PHASE_FENCE();
    STAMP(P, 3);

    {   // E: ph = Lq^-1 p ; [cp ; py] = Qf^T ph
        double ph[SN];
#pragma unroll
        for (int s = 0; s < SN; ++s) ph[s] = p0[s];
        tri_solve<SN, N>(C::SPLIT ? relabel(lds + C::oLq) : lds + C::oLq, st.rdq, ph, r);
        if (E > 0) {
            apply_QfT<C>(lds, st.tau, ph, r);
            shift_down<C>(ph, st.py, r);
        }
        // cp = first R entries
#pragma unroll
        for (int s = 0; s < SN; ++s) {
            const int i = r + 16 * s;
            double *dst = i < R ? lds + C::oCp + i : dummy;
            *dst = ph[s];
        }
    }
    {   // F: h' = h - W xy ;  w1 = W^T 1
        double hp[SM];
#pragma unroll
        for (int s = 0; s < SM; ++s) hp[s] = h0[s];
        if (E > 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double xb = BC(st.xy, e);
#pragma unroll
                for (int s = 0; s < SM; ++s) hp[s] = fma(-st.Gh[s][R + e], xb, hp[s]);
            }
            double one[SM];
#pragma unroll
            for (int s = 0; s < SM; ++s) one[s] = (r + 16 * s < M) ? 1.0 : 0.0;
            mul_WT<C>(st, one, st.w1, r);
        }
        vec_put<SM>(lds + C::oHp, hp, M, r, dummy);
    }
    __builtin_amdgcn_sched_barrier(0);
    DQP_PHASE_FENCE();
    STAMP(P, 4);

    if (C::EARLY && !C::PARK) park_GU<C>(P.workspace + qp * (long long)C::wsQP, st, r);
    if (E > 0) {
        double *ws = P.workspace + qp * (long long)C::wsQP;
#pragma unroll
        for (int s = 0; s < SE; ++s) {
            const int k = r + 16 * s;
            if (k < E) { ws[C::wXy + k] = st.xy[s]; ws[C::wPy + k] = st.py[s]; ws[C::wW1 + k] = st.w1[s]; }
        }
    }
    if (C::PARKWU && E > 0) {
        // three register slots per N-space matrix: W and U leave the register file here (phase G
        // needs the room for Lq Qf) and come back for the epilogue
        if (C::PARK || !C::EARLY) park_GU<C>(P.workspace + qp * (long long)C::wsQP, st, r);
#pragma unroll
        for (int s = 0; s < SM; ++s)
#pragma unroll
            for (int e = 0; e < E; ++e) st.Gh[s][R + e] = 0.0;
#pragma unroll
        for (int s = 0; s < SE; ++s)
#pragma unroll
            for (int c = 0; c < N; ++c) st.Ah[s][c] = 0.0;
        __builtin_amdgcn_sched_barrier(0);
        DQP_PHASE_FENCE();
    }
    {   // G: LqZ = (Lq Qf)[:, :R] ;  qv = (Lq Qf)[:, R:] w1
        double Ld[SN][N];
        const double *Lp = lds + C::oLq;
#pragma unroll
        for (int s = 0; s < SN; ++s) {
            const int i = r + 16 * s, ic = i < N ? i : N - 1;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j > 16 * s + 15) { Ld[s][j] = 0.0; continue; }
                const double v = Lp[tri(ic) + (j <= ic ? j : 0)];
                Ld[s][j] = (j <= i && i < N) ? v : 0.0;
            }
        }
        if (E > 0) {
#pragma unroll
            for (int k = E - 1; k >= 0; --k) {
                const int ck = R + k;
                const double *tl = lds + C::oTl + C::toff(k);
                const double tauk = BC(st.tau, k);
                double w[SN];
#pragma unroll
                for (int s = 0; s < SN; ++s) w[s] = Ld[s][ck];
#pragma unroll
                for (int c = 0; c < ck; ++c) {
                    const double uc = tl[c];
#pragma unroll
                    for (int s = 0; s < SN; ++s) w[s] = fma(Ld[s][c], uc, w[s]);
                }
#pragma unroll
                for (int s = 0; s < SN; ++s) { w[s] *= tauk; Ld[s][ck] -= w[s]; }
#pragma unroll
                for (int c = 0; c < ck; ++c) {
                    const double uc = tl[c];
#pragma unroll
                    for (int s = 0; s < SN; ++s) Ld[s][c] = fma(-w[s], uc, Ld[s][c]);
                }
                if (C::PIN) {
#pragma unroll
                    for (int s = 0; s < SN; ++s) { pin(Ld[s][0]); pin(Ld[s][ck]); }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < SN; ++s)
#pragma unroll
            for (int t = 0; t < R; ++t) st.LqZ[s][t] = Ld[s][t];
        double qv[SN];
#pragma unroll
        for (int s = 0; s < SN; ++s) qv[s] = 0.0;
        if (E > 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double wb = BC(st.w1, e);
#pragma unroll
                for (int s = 0; s < SN; ++s) qv[s] = fma(Ld[s][R + e], wb, qv[s]);
            }
        }
        vec_put<SN>(lds + C::oQv, qv, N, r, dummy);
    }
    __builtin_amdgcn_sched_barrier(0);
    DQP_PHASE_FENCE();
    STAMP(P, 5);

    // H: the reflector tails are done for now: park them in the caller's workspace (read back
    // for the epilogue), then Rm = Gz Gz^T goes, packed, into the same LDS region.
    {
        if (!C::EARLY) {
            double *ws = P.workspace + qp * (long long)C::wsQP;
            for (int e = r; e < C::tailsz; e += 16) ws[C::wTl + e] = lds[C::oTl + e];
            for (int e = r; e < tri(N); e += 16) ws[C::wLq + e] = lds[C::oLq + e];
            if (!C::PARKWU) park_GU<C>(ws, st, r);
#pragma unroll
            for (int s = 0; s < SN; ++s)
                if (r + 16 * s < N) ws[C::wRdq + r + 16 * s] = st.rdq[s];
        }
    }
    __syncthreads();
    {
        double *Rp = lds + C::oR;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int sj = j >> 4, lj = j & 15;
            double acc[SM];
#pragma unroll
            for (int s = 0; s < SM; ++s) acc[s] = 0.0;
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const double gb = rb(st.Gh[sj][c], lj);
#pragma unroll
                for (int s = 0; s < SM; ++s)
                    if (16 * s + 15 >= j) acc[s] = fma(st.Gh[s][c], gb, acc[s]);
            }
#pragma unroll
            for (int s = 0; s < SM; ++s) {
                if (16 * s + 15 < j) continue;
                const int i = r + 16 * s;
                double *dst = (i < M && j <= i) ? Rp + tri(i) + j : dummy;
                *dst = acc[s];
            }
        }
    }
    __syncthreads();   // factor_T reads R[j][i] written by the lane that owns row j
#pragma unroll
    for (int s = 0; s < SM; ++s) {
        const int i = r + 16 * s;
        const double v = (lds + C::oR)[tri(i < M ? i : M - 1) + (i < M ? i : M - 1)];
        st.rdiag[s] = i < M ? v : 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
    DQP_PHASE_FENCE();
}

// y = LqZ v + cc * qv   (N-space), returns sum of squares partial (lane-local)
template <class C>
__device__ __forceinline__ double rx_norm2_partial(const State<C> &st, const double *lds,
                                                   const double (&rw)[C::SR], double cc, int r)
{
    double rx[C::SN];
    vec_get<C::SN>(lds + C::oQv, rx, C::N, r);
#pragma unroll
    for (int s = 0; s < C::SN; ++s) rx[s] *= cc;
#pragma unroll
    for (int t = 0; t < C::R; ++t) {
        const double vb = BC(rw, t);
#pragma unroll
        for (int s = 0; s < C::SN; ++s) rx[s] = fma(st.LqZ[s][t], vb, rx[s]);
    }
    double a = 0.0;
#pragma unroll
    for (int s = 0; s < C::SN; ++s) a = fma(rx[s], rx[s], a);
    return a;
}

// Back to the caller's coordinates from the best iterate parked in LDS (oBw, oBs, oBz):
//   x = Lq^-T Qf [w* ; xy],  y = U^-T (c* delta w1 - xy - py - W^T z*);  writes zhat, lam, slack, nu,
// info, best_resid.  Shared by the forward kernel and the batch rule's finish pass.
template <class C>
__device__ __forceinline__ void epilogue(const KParams &P, long long qp, int r, double *lds, State<C> &st,
                                         double bestc, double delta, int iters, double best, bool live,
                                         bool reload_tails)
{
    constexpr int N = C::N, M = C::M, E = C::E, R = C::R;
    constexpr int SN = C::SN, SM = C::SM, SE = C::SE, SR = C::SR;
    bool inM[SM];
#pragma unroll
    for (int s = 0; s < SM; ++s) inM[s] = r + 16 * s < M;
    // the reflector tails come back from the workspace into the (now idle) Gz Gz^T region
    __syncthreads();
    if (E > 0 && reload_tails) {
        const double *ws = P.workspace + qp * (long long)C::wsQP + C::wTl;
        for (int e = r; e < C::tailsz; e += 16) lds[C::oTl + e] = ws[e];
    }
    __syncthreads();

    // recover x = Lq^-T Qf [w* ; xy],  y = U^-T (c* delta w1 - xy - py - W^T z*)
    double bw[SR], bs[SM], bz[SM];
    vec_get<SR>(lds + C::oBw, bw, R, r);
    vec_get<SM>(lds + C::oBs, bs, M, r);
    vec_get<SM>(lds + C::oBz, bz, M, r);
    double xh[SN];
#pragma unroll
    for (int s = 0; s < SN; ++s) xh[s] = (s < SR && r + 16 * s < R) ? bw[s < SR ? s : 0] : 0.0;
    double yv[SE];
#pragma unroll
    for (int s = 0; s < SE; ++s) yv[s] = 0.0;
    if (E > 0) {
        if (C::PARKWU) unpark_WU<C>(P.workspace + qp * (long long)C::wsQP, st, r);
        shift_up<C>(st.xy, xh, r);
        apply_Qf<C>(lds, st.tau, xh, r);
        double wz[SE];
        mul_WT<C>(st, bz, wz, r);
#pragma unroll
        for (int s = 0; s < SE; ++s)
            yv[s] = (r + 16 * s < E) ? bestc * delta * st.w1[s] - st.xy[s] - st.py[s] - wz[s] : 0.0;
        solve_UT<C>(st, yv, r);
    }
    tri_solve_T<SN, N>(lds + C::oLq, st.rdq, xh, r);
    if (live) {
#pragma unroll
        for (int s = 0; s < SN; ++s)
            if (r + 16 * s < N) P.zhat[qp * N + r + 16 * s] = xh[s];
#pragma unroll
        for (int s = 0; s < SM; ++s)
            if (inM[s]) { P.lam[qp * M + r + 16 * s] = bz[s]; P.slack[qp * M + r + 16 * s] = bs[s]; }
        if (E > 0) {
#pragma unroll
            for (int s = 0; s < SE; ++s)
                if (r + 16 * s < E) P.nu[qp * E + r + 16 * s] = yv[s];
        }
        if (r == 0) {
            if (P.info) { P.info[2 * qp] = st.status; P.info[2 * qp + 1] = iters; }
            if (P.best_resid) P.best_resid[qp] = best;
        }
    }
}

template <class C>
__global__ __launch_bounds__(64) void forward_kernel(KParams P)
{
    constexpr int N = C::N, M = C::M, E = C::E, R = C::R;
    constexpr int SN = C::SN, SM = C::SM, SE = C::SE, SR = C::SR;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x, r = lane & 15, qrow = lane >> 4;
    long long qp = (long long)blockIdx.x * 4 + qrow;
    bool live = qp < P.B;
    if (!live) qp = P.B - 1;                     // duplicate the last QP; its stores are masked
    int maxIter = P.maxIter;
    const bool batch = (P.flags & DQP_FLAG_BATCH_TERMINATION) != 0;
    const bool strict = (P.flags & DQP_FLAG_STRICT_GET_STEP) != 0;
    if (P.cap) {        // pass 2 of the batch rule: only the listed QPs, up to the reference's stop
        maxIter = min(maxIter, P.cap[0]);
        live = live && term_flagged(P, qp);
        if (__builtin_amdgcn_ballot_w64(live) == 0) return;
    }
    double *lds = sm + qrow * C::ldsQPpad;
    double *dummy = lds + C::oDummy + r;

    term_zero_acc(P);
    State<C> st;
    STAMP(P, 0);
    setup<C>(P, qp, r, lds, st);
    STAMP(P, 15);

    bool inM[SM], inR[SR];
#pragma unroll
    for (int s = 0; s < SM; ++s) inM[s] = r + 16 * s < M;
#pragma unroll
    for (int s = 0; s < SR; ++s) inR[s] = r + 16 * s < R;

    double T[SM][M], rdu[SM];
    double w[SR], s_[SM], z[SM];
    double cc = 1.0, delta = 0.0;
    {   // initial point, d = 1 (batch.py:60-86):  (Gz Gz^T + I) z = -h' - Gz cp,  w = -cp - Gz^T z
        double one[SM], hp[SM], cp[SR], g[SM], gtz[SR];
#pragma unroll
        for (int s = 0; s < SM; ++s) one[s] = inM[s] ? 1.0 : 0.0;
        vec_get<SM>(lds + C::oHp, hp, M, r);
        vec_get<SR>(lds + C::oCp, cp, R, r);
        mul_Gz<C>(st, cp, g);
#pragma unroll
        for (int s = 0; s < SM; ++s) g[s] = inM[s] ? -hp[s] - g[s] : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        factor_T<C>(lds, T, st.rdiag, one, rdu, r);
        lu_solve<SM, M>(T, rdu, g, r);
        __builtin_amdgcn_sched_barrier(0);
        mul_GzT<C>(st, g, gtz, r);
#pragma unroll
        for (int s = 0; s < SR; ++s) w[s] = inR[s] ? -cp[s] - gtz[s] : 0.0;
        double ms = INFINITY, mz = INFINITY;
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            z[s] = inM[s] ? g[s] : 0.0;
            s_[s] = -z[s];
            ms = fmin(ms, inM[s] ? s_[s] : INFINITY);
            mz = fmin(mz, inM[s] ? z[s] : INFINITY);
        }
        ms = row_min(ms); mz = row_min(mz);                                 // batch.py:76-86
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            if (ms < 0.0 && inM[s]) s_[s] -= ms - 1.0;
            if (mz < 0.0 && inM[s]) z[s] -= mz - 1.0;
        }
        delta = mz < 0.0 ? 1.0 - mz : 0.0;        // rho_0 = delta * W^T 1
        if (r == 0) P.workspace[qp * (long long)C::wsQP + C::wDl] = delta;
    }

    // gz = Gz^T z is carried incrementally (updated with the step's own transposed product)
    double gz[SR];
    mul_GzT<C>(st, z, gz, r);
    STAMP(P, 6);
    double best = INFINITY, bestc = 1.0;
    bool have_best = false, done = false;
    int nNot = 0, iters = 0;

    for (int it = 0; it < maxIter; ++it) {
        // residuals                                                        batch.py:93-108
        double rw[SR], rz[SM];
        {
            double cp[SR], hp[SM], gw[SM];
            vec_get<SR>(lds + C::oCp, cp, R, r);
            vec_get<SM>(lds + C::oHp, hp, M, r);
            mul_Gz<C>(st, w, gw);
#pragma unroll
            for (int s = 0; s < SR; ++s) rw[s] = inR[s] ? w[s] + cp[s] + gz[s] : 0.0;
#pragma unroll
            for (int s = 0; s < SM; ++s) rz[s] = inM[s] ? gw[s] + s_[s] - hp[s] : 0.0;
        }
        double sz = 0.0, nz2 = 0.0;
#pragma unroll
        for (int s = 0; s < SM; ++s) { sz = fma(s_[s], z[s], sz); nz2 = fma(rz[s], rz[s], nz2); }
        double nx2 = rx_norm2_partial<C>(st, lds, rw, cc * delta, r);
        sz = row_sum(sz); nz2 = row_sum(nz2); nx2 = row_sum(nx2);
        const double mu = fabs(sz * (1.0 / M));
        const double resid = sqrt(nz2) + sqrt(nx2) + M * mu;
        // best-iterate tracking / per-problem termination (uniform inside a DPP row)
        if (!done) {
            iters = it + 1;
            if (!have_best || resid < best) {
                nNot = 0; have_best = true; best = resid; bestc = cc;
                vec_put<SR>(lds + C::oBw, w, R, r, dummy);
                vec_put<SM>(lds + C::oBs, s_, M, r, dummy);
                vec_put<SM>(lds + C::oBz, z, M, r, dummy);
                if (P.snap && live) {        // batch rule: the finish pass picks the best iterate up to I*
                    double *sp = P.snap + ((long long)it * P.B + qp) * C::snapDim;
#pragma unroll
                    for (int s = 0; s < SR; ++s)
                        if (inR[s]) sp[r + 16 * s] = w[s];
#pragma unroll
                    for (int s = 0; s < SM; ++s)
                        if (inM[s]) { sp[R + r + 16 * s] = s_[s]; sp[R + M + r + 16 * s] = z[s]; }
                    if (r == 0) sp[R + 2 * M] = cc;
                }
            } else {
                nNot += 1;
            }
            if (batch) {                       // the stop is decided over the batch (dqp_term.hip)
                if (P.hist && r == 0 && live) hist_put(P, qp, it, resid, mu);
                done = !(fabs(resid) < INFINITY);
            } else if ((nNot >= P.notImprovedLim && best < P.stallTol) || best < P.eps || mu > 1e32 ||
                       !(fabs(resid) < INFINITY))
                done = true;
        }
        // the wave leaves when all four of its QPs are done
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
        if (it == 1) STAMP(P, 9);

        // phase 1 (Gz live, T dead): affine right-hand side  g = rz - rs/d - Gz rw,  rs/d = s
        double dza[SM], dsa[SM];
        {
            double gu[SM];
            mul_Gz<C>(st, rw, gu);
#pragma unroll
            for (int s = 0; s < SM; ++s) dza[s] = rz[s] - s_[s] - gu[s];
        }
        __builtin_amdgcn_sched_barrier(0);
        // phase 2 (T live): factor + affine and corrector solves              batch.py:110-181
        double dinv[SM];
#pragma unroll
        for (int s = 0; s < SM; ++s) dinv[s] = inM[s] ? s_[s] * frcp(z[s]) : 0.0;   // 1/d, d = z/s
        factor_T<C>(lds, T, st.rdiag, dinv, rdu, r);
        if (it == 1) STAMP(P, 10);
        lu_solve<SM, M>(T, rdu, dza, r);
        if (it == 1) STAMP(P, 11);
        double rzv[SM], rsv[SM];           // formed only now: not live across the factorisation
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            rzv[s] = inM[s] ? frcp(z[s]) : 0.0;
            rsv[s] = inM[s] ? frcp(s_[s]) : 0.0;
        }
        double tm = 0.0;
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            dsa[s] = (-z[s] - dza[s]) * dinv[s];
            tm = fmax(tm, fmax(-dza[s] * rzv[s], -dsa[s] * rsv[s]));
        }
        // DQP_FLAG_STRICT_GET_STEP: batch.py:211-214 divides by the step; an exactly-zero component turns the
        // reference's iterate NaN and the problem keeps the best iterate it had
        bool zero_step = false;
        if (strict) {
#pragma unroll
            for (int s = 0; s < SM; ++s) zero_step |= inM[s] && (dza[s] == 0.0 || dsa[s] == 0.0);
        }
        double alpha = frcp(fmax(row_max(tm), 1.0));                           // min(step, 1)
        double t3 = 0.0;
#pragma unroll
        for (int s = 0; s < SM; ++s)
            t3 = fma(fma(alpha, dsa[s], s_[s]), fma(alpha, dza[s], z[s]), t3);
        t3 = row_sum(t3);
        double sig = t3 / sz;
        sig = sig * sig * sig;
        // corrector: rw = rz = 0, rs = (-mu sig + ds_a dz_a)/s                batch.py:171-181
        double rsc[SM], dzc[SM], dz[SM], ds[SM];
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            rsc[s] = fma(dsa[s], dza[s], -mu * sig) * rsv[s];
            dzc[s] = -rsc[s] * dinv[s];
        }
        lu_solve<SM, M>(T, rdu, dzc, r);
        if (it == 1) STAMP(P, 12);
        tm = 0.0;
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            dz[s] = dza[s] + dzc[s];
            ds[s] = fma(-rsc[s] - dzc[s], dinv[s], dsa[s]);
            tm = fmax(tm, fmax(-dz[s] * rzv[s], -ds[s] * rsv[s]));
        }
        if (strict) {
#pragma unroll
            for (int s = 0; s < SM; ++s) zero_step |= inM[s] && (dz[s] == 0.0 || ds[s] == 0.0);
            if ((__builtin_amdgcn_ballot_w64(zero_step) >> (lane & 48)) & 0xffffull) done = true;
        }
        __builtin_amdgcn_sched_barrier(0);
        // phase 3 (Gz live, T dead): dw = -rw - Gz^T dz
        double gtd[SR];
        mul_GzT<C>(st, dz, gtd, r);
        alpha = frcp(fmax(row_max(tm) * (1.0 / 0.999), 1.0));                  // min(0.999 step, 1)
        if (!done) {
#pragma unroll
            for (int s = 0; s < SR; ++s) {
                w[s] = fma(alpha, -rw[s] - gtd[s], w[s]);
                gz[s] = fma(alpha, gtd[s], gz[s]);
            }
#pragma unroll
            for (int s = 0; s < SM; ++s) { s_[s] = fma(alpha, ds[s], s_[s]); z[s] = fma(alpha, dz[s], z[s]); }
            cc *= (1.0 - alpha);
        }
        if (it == 1) STAMP(P, 13);
        if (it == 0) STAMP(P, 8);
    }
    STAMP(P, 7);

    if (P.hist && r == 0 && live) hist_fill(P, qp, iters);
    epilogue<C>(P, qp, r, lds, st, bestc, delta, iters, best, live, true);
    STAMP(P, 14);
}

// The factorisation context a forward launch left in the workspace -> LDS (Lq, reflector tails) and
// registers ([Gz | W], U, tau, 1/diag(U), 1/diag(Lq)).
template <class C>
__device__ __forceinline__ void load_ctx(const KParams &P, long long qp, int r, double *lds, State<C> &st)
{
    constexpr int N = C::N, M = C::M, E = C::E, R = C::R;
    constexpr int SN = C::SN, SM = C::SM, SE = C::SE;
    st.status = DQP_STATUS_OK;
    {
        const double *ws = P.workspace + qp * (long long)C::wsQP;
        for (int e = r; e < C::tailsz; e += 16) lds[C::oTl + e] = ws[C::wTl + e];
        for (int e = r; e < tri(N); e += 16) lds[C::oLq + e] = ws[C::wLq + e];
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            const int i = r + 16 * s, ic = i < M ? i : M - 1;
#pragma unroll
            for (int c = 0; c < N; ++c) {
                const double v = ws[C::wG + c * M + ic];
                st.Gh[s][c] = (16 * s + 15 >= M) ? mask_hi(v, i < M) : v;
            }
        }
#pragma unroll
        for (int s = 0; s < SE; ++s) {
            const int k = r + 16 * s, kc = k < E ? k : E - 1;
#pragma unroll
            for (int c = 0; c < N; ++c) st.Ah[s][c] = 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double v = ws[C::wU + e * E + kc];
                st.Ah[s][R + e] = k < E ? v : 0.0;
            }
            const double t = ws[C::wTau + kc], u = ws[C::wRdu + kc];
            st.tau[s] = k < E ? t : 0.0;
            st.rdu1[s] = k < E ? u : 0.0;
        }
#pragma unroll
        for (int s = 0; s < SN; ++s) {
            const int i = r + 16 * s;
            const double v = ws[C::wRdq + (i < N ? i : N - 1)];
            st.rdq[s] = i < N ? v : 0.0;
        }
    }
}

// Pass 2 of the batch-coupled rule when pass 1 kept its improving iterates (P.snap): nothing is
// solved again.  A flagged problem (its best iterate came at or after the batch's stop I*) takes the
// best of its iterates before I* from the snapshots, the factorisation context from the workspace,
// and runs the epilogue only.
template <class C>
__global__ __launch_bounds__(64) void finish_kernel(KParams P)
{
    constexpr int M = C::M, E = C::E, R = C::R;
    constexpr int SM = C::SM, SE = C::SE, SR = C::SR;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x, r = lane & 15, qrow = lane >> 4;
    long long qp = (long long)blockIdx.x * 4 + qrow;
    bool live = qp < P.B;
    if (!live) qp = P.B - 1;
    live = live && term_flagged(P, qp);
    if (__builtin_amdgcn_ballot_w64(live) == 0) return;
    const int cap = min(P.maxIter, P.cap[0]);
    double *lds = sm + qrow * C::ldsQPpad;
    double *dummy = lds + C::oDummy + r;
    State<C> st;
    load_ctx<C>(P, qp, r, lds, st);
    const double *ws = P.workspace + qp * (long long)C::wsQP;
#pragma unroll
    for (int s = 0; s < SE; ++s) {
        const int k = r + 16 * s, kc = k < E ? k : (E > 0 ? E - 1 : 0);
        const double a = ws[C::wXy + kc], b = ws[C::wPy + kc], c = ws[C::wW1 + kc];
        st.xy[s] = (E > 0 && k < E) ? a : 0.0;
        st.py[s] = (E > 0 && k < E) ? b : 0.0;
        st.w1[s] = (E > 0 && k < E) ? c : 0.0;
    }
    const double delta = ws[C::wDl];
    // the problem's best iteration before the stop (same rule as the kernels: first strict minimum)
    const double2 *h = reinterpret_cast<const double2 *>(P.histIn) + qp;
    double best = h[0].x;
    int arg = 0;
    for (int it = 1; it < cap; ++it) {
        const double v = h[(long long)it * P.B].x;
        if (v < best) { best = v; arg = it; }
    }
    const double *sp = P.snap + ((long long)arg * P.B + qp) * C::snapDim;
    double bw[SR], bs[SM], bz[SM];
#pragma unroll
    for (int s = 0; s < SR; ++s) { const int i = r + 16 * s; const double v = sp[i < R ? i : 0]; bw[s] = i < R ? v : 0.0; }
#pragma unroll
    for (int s = 0; s < SM; ++s) {
        const int i = r + 16 * s, ic = i < M ? i : 0;
        const double a = sp[R + ic], b = sp[R + M + ic];
        bs[s] = i < M ? a : 0.0;
        bz[s] = i < M ? b : 0.0;
    }
    const double bestc = sp[R + 2 * M];
    vec_put<SR>(lds + C::oBw, bw, R, r, dummy);
    vec_put<SM>(lds + C::oBs, bs, M, r, dummy);
    vec_put<SM>(lds + C::oBz, bz, M, r, dummy);
    st.status = P.info ? P.info[2 * qp] : DQP_STATUS_OK;       // pass 1's status stands
    epilogue<C>(P, qp, r, lds, st, bestc, delta, cap, best, live, false);
}

// Backward pass in the same null-space coordinates (reference: qp.py:128-183 = factor_kkt +
// solve_kkt with rhs (dl_dzhat, 0, 0, 0) + the outer-product gradient formulas).  With
// gq = Qf^T Lq^-1 g split into its null-space part gz and range part gy:
//   (Gz Gz^T + diag(s/lam)) dlam = -Gz gz,   dw = -(gz + Gz^T dlam),   dx = Lq^-T Qf [dw ; 0],
//   dnu = -U^-T (gy + W^T dlam).
// Nothing is refactored: Lq, the reflectors, [Gz | W] and U come back from the workspace the
// forward kernel of the same (Q, G, A) filled (DQP_FLAG_BACKWARD_CTX), like the reference's
// ctx.Q_LU / S_LU / R; T is accumulated straight into registers.
template <class C>
__global__ __launch_bounds__(64) void backward_kernel(KParams P)
{
    constexpr int N = C::N, M = C::M, E = C::E, R = C::R;
    constexpr int SN = C::SN, SM = C::SM, SE = C::SE, SR = C::SR;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x, r = lane & 15, qrow = lane >> 4;
    long long qp = (long long)blockIdx.x * 4 + qrow;
    const bool live = qp < P.B;
    if (!live) qp = P.B - 1;
    double *lds = sm + qrow * C::ldsQPpad;

    State<C> st;
    load_ctx<C>(P, qp, r, lds, st);
    __syncthreads();

    bool inN[SN], inM[SM], inE[SE];
#pragma unroll
    for (int s = 0; s < SN; ++s) inN[s] = r + 16 * s < N;
#pragma unroll
    for (int s = 0; s < SM; ++s) inM[s] = r + 16 * s < M;
#pragma unroll
    for (int s = 0; s < SE; ++s) inE[s] = r + 16 * s < E;

    double zh[SN], g[SN], lam[SM], dinv[SM], nu[SE];
#pragma unroll
    for (int s = 0; s < SN; ++s) {
        zh[s] = inN[s] ? P.zin[qp * N + r + 16 * s] : 0.0;
        g[s] = inN[s] ? P.gin[qp * N + r + 16 * s] : 0.0;
    }
#pragma unroll
    for (int s = 0; s < SM; ++s) {
        lam[s] = inM[s] ? P.lamin[qp * M + r + 16 * s] : 0.0;
        const double sl = inM[s] ? P.slackin[qp * M + r + 16 * s] : 1.0;
        if (P.flags & DQP_FLAG_DENSE_BACKWARD) dinv[s] = inM[s] ? sl / lam[s] : 0.0;
        else dinv[s] = inM[s] ? fmax(sl, 1e-8) / fmax(lam[s], 1e-8) : 0.0;     // qp.py:149
    }
#pragma unroll
    for (int s = 0; s < SE; ++s) nu[s] = (E > 0 && inE[s]) ? P.nuin[qp * E + r + 16 * s] : 0.0;

    // gq = Qf^T Lq^-1 g
    tri_solve<SN, N>(lds + C::oLq, st.rdq, g, r);
    double gy[SE];
#pragma unroll
    for (int s = 0; s < SE; ++s) gy[s] = 0.0;
    if (E > 0) {
        apply_QfT<C>(lds, st.tau, g, r);
        shift_down<C>(g, gy, r);
    }
    double gz[SR];
#pragma unroll
    for (int s = 0; s < SR; ++s) gz[s] = (r + 16 * s < R) ? g[s] : 0.0;

    double dlam[SM];
    mul_Gz<C>(st, gz, dlam);
#pragma unroll
    for (int s = 0; s < SM; ++s) dlam[s] = -dlam[s];
    __builtin_amdgcn_sched_barrier(0);
    {
        double T[SM][M], rdu[SM];
#pragma unroll
        for (int j = 0; j < M; ++j) {                  // column by column: short live ranges
            const int sj = j >> 4, lj = j & 15;
            double acc[SM];
#pragma unroll
            for (int s = 0; s < SM; ++s) acc[s] = (r + 16 * s == j) ? dinv[s] : 0.0;
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const double gb = rb(st.Gh[sj][c], lj);
#pragma unroll
                for (int s = 0; s < SM; ++s) acc[s] = fma(st.Gh[s][c], gb, acc[s]);
            }
#pragma unroll
            for (int s = 0; s < SM; ++s) T[s][j] = acc[s];
            __builtin_amdgcn_sched_barrier(0);
        }
        lu_rows<SM, M>(T, rdu, r);
        lu_solve<SM, M>(T, rdu, dlam, r);
    }
    __builtin_amdgcn_sched_barrier(0);
    double dxh[SN], dnu[SE];
    {
        double gtd[SR];
        mul_GzT<C>(st, dlam, gtd, r);
#pragma unroll
        for (int s = 0; s < SN; ++s) dxh[s] = (s < SR && r + 16 * s < R) ? -(gz[s < SR ? s : 0] + gtd[s < SR ? s : 0]) : 0.0;
    }
#pragma unroll
    for (int s = 0; s < SE; ++s) dnu[s] = 0.0;
    if (E > 0) {
        apply_Qf<C>(lds, st.tau, dxh, r);
        double wz[SE];
        mul_WT<C>(st, dlam, wz, r);
#pragma unroll
        for (int s = 0; s < SE; ++s) dnu[s] = inE[s] ? -(gy[s] + wz[s]) : 0.0;
        solve_UT<C>(st, dnu, r);
    }
    tri_solve_T<SN, N>(lds + C::oLq, st.rdq, dxh, r);                   // dx = Lq^-T dxh

    if (!live) return;
    if (P.mdc || P.mdC || P.mdF || P.mdf || P.mdx0) {
        // MPC-structured gradients: the adjoint of the assembly (qp_wrapper.py:638-679) applied on
        // chip -- only the diagonal blocks of dQ = sym(dx z^T), the F-blocks of dA = dnu z^T + nu dx^T,
        // df = -db[dynamics rows] = dnu, dx0 = db[initial rows] = -dnu, dc = dp = dx.
        const int n = P.mn, m = P.mm, nt = n + m, T = P.mT;
        const long long Bq = P.B;
        int tl[SN], jl[SN];                     // knot / offset of the lane's own columns
#pragma unroll
        for (int s = 0; s < SN; ++s) { const int c = r + 16 * s; tl[s] = c / nt; jl[s] = c - tl[s] * nt; }
        if (P.mdc) {
#pragma unroll
            for (int s = 0; s < SN; ++s)
                if (inN[s]) P.mdc[((long long)tl[s] * Bq + qp) * nt + jl[s]] = dxh[s];
        }
        if (P.mdC) {
            int ti = 0, ji = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double dxi = BC(dxh, i), zi = BC(zh, i);
#pragma unroll
                for (int s = 0; s < SN; ++s)
                    if (inN[s] && tl[s] == ti)
                        P.mdC[(((long long)ti * Bq + qp) * nt + ji) * nt + jl[s]] = 0.5 * (dxi * zh[s] + zi * dxh[s]);
                if (++ji == nt) { ji = 0; ++ti; }
            }
        }
        if (E > 0) {
            int te = 0, je = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double dni = BC(dnu, e), ni = BC(nu, e);
                if (te < T - 1) {
                    if (P.mdF) {
#pragma unroll
                        for (int s = 0; s < SN; ++s)
                            if (inN[s] && tl[s] == te)
                                P.mdF[(((long long)te * Bq + qp) * n + je) * nt + jl[s]] = dni * zh[s] + ni * dxh[s];
                    }
                    if (P.mdf && r == 0) P.mdf[((long long)te * Bq + qp) * n + je] = dni;
                } else if (P.mdx0 && r == 0) {
                    P.mdx0[qp * n + je] = -dni;
                }
                if (++je == n) { je = 0; ++te; }
            }
        }
        if (r == 0 && P.info) { P.info[2 * qp] = st.status; P.info[2 * qp + 1] = 0; }
        return;
    }
    // gradients (qp.py:158-181); each lane writes its own rows
#pragma unroll
    for (int s = 0; s < SN; ++s)
        if (P.dp && inN[s]) P.dp[qp * N + r + 16 * s] = dxh[s];
#pragma unroll
    for (int s = 0; s < SM; ++s)
        if (P.dh && inM[s]) P.dh[qp * M + r + 16 * s] = -dlam[s];
    if (E > 0) {
#pragma unroll
        for (int s = 0; s < SE; ++s)
            if (P.db && inE[s]) P.db[qp * E + r + 16 * s] = -dnu[s];
    }
    // Outer products with lanes along the contiguous column axis: for row i the 16 lanes of the
    // QP's DPP row write 16 consecutive doubles (one 128-byte segment) per slot.
    if (P.dQ) {
        double *o = P.dQ + qp * N * N;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double dxi = BC(dxh, i), zi = BC(zh, i);
#pragma unroll
            for (int s = 0; s < SN; ++s)
                if (inN[s]) o[i * N + r + 16 * s] = 0.5 * (dxi * zh[s] + zi * dxh[s]);
        }
    }
    if (P.dG) {
        double *o = P.dG + qp * M * N;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const double dli = BC(dlam, i), li = BC(lam, i);
#pragma unroll
            for (int s = 0; s < SN; ++s)
                if (inN[s]) o[i * N + r + 16 * s] = dli * zh[s] + li * dxh[s];
        }
    }
    if (P.dA && E > 0) {
        double *o = P.dA + qp * E * N;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const double dni = BC(dnu, i), ni = BC(nu, i);
#pragma unroll
            for (int s = 0; s < SN; ++s)
                if (inN[s]) o[i * N + r + 16 * s] = dni * zh[s] + ni * dxh[s];
        }
    }
    if (r == 0 && P.info) { P.info[2 * qp] = st.status; P.info[2 * qp + 1] = 0; }
}

template <class C, class K>
int launch(K kernel, const KParams &P, void *stream)
{
    const size_t lds = (size_t)4 * C::ldsQPpad * sizeof(double);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DQP_ERR_LAUNCH;
    const int blocks = (P.B + 3) / 4;
    DQP_LAUNCH(kernel, dim3(blocks), dim3(64), lds, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // namespace r16n

#if !defined(DQP_R16_N) || !defined(DQP_R16_M) || !defined(DQP_R16_E)
#error "compile with -DDQP_R16_N=.. -DDQP_R16_M=.. -DDQP_R16_E=.."
#endif
#define DQP_CAT2(a, n, m, e) a##n##_##m##_##e
#define DQP_CAT(a, n, m, e) DQP_CAT2(a, n, m, e)

#ifndef DQP_R16_BWD
int DQP_CAT(r16n_forward_, DQP_R16_N, DQP_R16_M, DQP_R16_E)(const KParams &P, void *stream)
{
    using C = r16n::Cfg<DQP_R16_N, DQP_R16_M, DQP_R16_E>;
    if (P.cap && P.snap && P.histIn) return r16n::launch<C>(r16n::finish_kernel<C>, P, stream);
    return r16n::launch<C>(r16n::forward_kernel<C>, P, stream);
}
#else
int DQP_CAT(r16n_backward_, DQP_R16_N, DQP_R16_M, DQP_R16_E)(const KParams &P, void *stream)
{
    using C = r16n::Cfg<DQP_R16_N, DQP_R16_M, DQP_R16_E>;
    return r16n::launch<C>(r16n::backward_kernel<C>, P, stream);
}
#endif

}  // namespace dqp
