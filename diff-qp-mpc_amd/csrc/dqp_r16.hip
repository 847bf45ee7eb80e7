// dqp_r16.hip -- DPP-row kernels: FOUR QPs per wavefront, one per 16-lane DPP row.
//
// For problems with nz, nineq <= 32 and neq <= 16 (compile-time sizes) every matrix of the
// interior-point iteration is held in REGISTERS, "row-distributed": row i of a matrix lives on
// lane (i % 16) of the QP's DPP row, in slot (i / 16), with all of its columns in consecutive
// VGPRs.  A vector is distributed the same way (element i on lane i%16, slot i/16).
// The only cross-lane primitive is the CDNA DPP row broadcast (`v_mov_b64_dpp row_newbcast:k`,
// full rate, no LDS, no SGPR round trip), so
//   * y = M x       is   for j: y[s] += M[s][j] * bcast_j(x)            (no reduction)
//   * LU / Cholesky / triangular solves are rank-1 updates with bcast of the pivot row/entry
//   * y = M^T v     is a lane-local partial product + a 4-round mirror butterfly (reduce-scatter)
// and four independent QPs advance in lock-step in the same instruction stream, which amortises
// every latency-bound dependent chain (pivots, substitution steps) over four problems.
// LDS only holds the per-QP constants that are touched once per iteration (packed lower
// triangles of R, Lq, L1: 8.4 KB per QP for the metric size).
//
// Math: identical to dqp_pdipm.hip (see its header): hat coordinates xh = Lq^T x, orthonormal
// equality rows At, T = R + diag(s/z) factored per iteration -- here as an unpivoted LU (what
// the reference itself does on GPUs, batch.py:8-19), because row-distributed storage gives
// the U rows needed for back-substitution for free.
//
// Reference functions covered: as dqp_pdipm.hip.

#include "dqp_r16_prims.h"

namespace dqp {
namespace r16 {

// ------------------------------------------------------------------ per-size configuration
template <int N_, int M_, int E_> struct Cfg {
    static constexpr int N = N_, M = M_, E = E_;
    static constexpr int SN = slots(N_), SM = slots(M_), SE = E_ > 0 ? slots(E_) : 1;
    static constexpr int EC = E_ > 0 ? E_ : 1;            // column count for E-sized arrays
    // packed triangles in LDS, per QP (doubles)
    static constexpr int oR = 0, oLq = tri(M_), oL1 = tri(M_) + tri(N_);
    // distributed vectors parked in LDS (element i at offset + i): p^, h, b~, best iterate
    static constexpr int oPh = tri(M_) + tri(N_) + tri(E_);
    static constexpr int oH = oPh + N_, oBt = oH + M_;
    static constexpr int oBx = oBt + E_, oBs = oBx + N_, oBz = oBs + M_, oBy = oBz + M_;
    static constexpr int oDummy = oBy + E_;                      // 16 write-only sink slots
    static constexpr int ldsQP = oDummy + 16;
    static constexpr int ldsQPpad = (ldsQP + 1) & ~1;
};

// Everything the iteration needs, produced by setup():
template <class C> struct State {
    double Gh[C::SM][C::N];      // Gh = G Lq^-T
    double Ah[C::SE][C::N];      // At = L1^-1 A Lq^-T (orthonormal rows)
    double rdq[C::SN];           // 1 / diag(Lq)
    double rd1[C::SE];           // 1 / diag(L1)
    double rdiag[C::SM];         // diag(R) of the lane's rows (factor_T rewrites the LDS copy)
    int status;
};

// One-time factorisations (reference: pre_factor_kkt, batch.py:377-428), all in registers.
// Writes packed R, Lq, L1 to this QP's LDS block.  Phases are separated by scheduling
// barriers so that the live register set of one phase is not extended into the next.
template <class C>
__device__ __forceinline__ void setup(const KParams &P, long long qp, int r, double *lds, State<C> &st)
{
    constexpr int N = C::N, M = C::M, E = C::E, SN = C::SN, SM = C::SM, SE = C::SE;
    double *dummy = lds + C::oDummy + r;
    st.status = DQP_STATUS_OK;
    STAMP(P, 0);
    {   // phase A: Q -> Lq (Cholesky) -> packed LDS
        double Lq[SN][N];
        load_rows<SN, N>(P.Q + qp * P.sQ, N, Lq, r);
        if (!chol_rows<SN, N>(Lq, st.rdq, r)) st.status = DQP_STATUS_Q_NOT_PD;
        tri_store<SN, N>(lds + C::oLq, Lq, r, dummy);
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    STAMP(P, 1);

    // phase B: rows of G and A:  row <- row Lq^-T, with Lq[j][k] read (row-uniformly) from LDS
    load_rows<SM, N>(P.G + qp * P.sG, M, st.Gh, r);
    if (E > 0) load_rows<SE, N>(P.A + qp * P.sA, E, st.Ah, r);
    {
        const double *Lp = lds + C::oLq;
#pragma unroll
        for (int j = 0; j < N; ++j) {
#pragma unroll
            for (int k = 0; k < j; ++k) {
                const double ljk = Lp[tri(j) + k];
#pragma unroll
                for (int s = 0; s < SM; ++s) st.Gh[s][j] = fma(-st.Gh[s][k], ljk, st.Gh[s][j]);
                if (E > 0) {
#pragma unroll
                    for (int s = 0; s < SE; ++s) st.Ah[s][j] = fma(-st.Ah[s][k], ljk, st.Ah[s][j]);
                }
            }
            const double rj = BC(st.rdq, j);
#pragma unroll
            for (int s = 0; s < SM; ++s) st.Gh[s][j] *= rj;
            if (E > 0) {
#pragma unroll
                for (int s = 0; s < SE; ++s) st.Ah[s][j] *= rj;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(P, 2);

    double W[SM][C::EC];
    if (E > 0) {
        constexpr int EC = C::EC;
        {   // phase C: S11 = Ah Ah^T = L1 L1^T ; At = L1^-1 Ah
            double L1[SE][EC];
#pragma unroll
            for (int j = 0; j < EC; ++j) {
                const int sj = j >> 4, lj = j & 15;
#pragma unroll
                for (int s = 0; s < SE; ++s) L1[s][j] = 0.0;
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    const double ab = rb(st.Ah[sj][c], lj);
#pragma unroll
                    for (int s = 0; s < SE; ++s) L1[s][j] = fma(st.Ah[s][c], ab, L1[s][j]);
                }
            }
            if (!chol_rows<SE, EC>(L1, st.rd1, r, 1e-13) && st.status == DQP_STATUS_OK)
                st.status = DQP_STATUS_A_RANK_DEF;
            // eliminate with unscaled rows, scale once at the end
#pragma unroll
            for (int g = 0; g < EC; ++g) {
                const int sg = g >> 4, lg = g & 15;
                const double rg = rb(st.rd1[sg], lg);
                double m[SE];
#pragma unroll
                for (int s = 0; s < SE; ++s) {
                    if (16 * s + 15 <= g) m[s] = 0.0;
                    else if (16 * s > g) m[s] = L1[s][g] * rg;
                    else m[s] = (r > lg) ? L1[s][g] * rg : 0.0;
                }
#pragma unroll
                for (int c = 0; c < N; ++c) {
                    const double ub = rb(st.Ah[sg][c], lg);
#pragma unroll
                    for (int s = 0; s < SE; ++s)
                        if (16 * s + 15 > g) st.Ah[s][c] = fma(-m[s], ub, st.Ah[s][c]);
                }
            }
#pragma unroll
            for (int s = 0; s < SE; ++s)
#pragma unroll
                for (int c = 0; c < N; ++c) st.Ah[s][c] *= st.rd1[s];
            tri_store<SE, EC>(lds + C::oL1, L1, r, dummy);
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(P, 3);
        // W = Gh At^T  (M x E)
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            const int se = e >> 4, le = e & 15;
#pragma unroll
            for (int s = 0; s < SM; ++s) W[s][e] = 0.0;
#pragma unroll
            for (int c = 0; c < N; ++c) {
                const double ab = rb(st.Ah[se][c], le);
#pragma unroll
                for (int s = 0; s < SM; ++s) W[s][e] = fma(st.Gh[s][c], ab, W[s][e]);
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < SE; ++s) st.rd1[s] = 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(P, 4);

    {   // phase D: R = Gh Gh^T - W W^T (the reference's own form, batch.py:399,420), packed -> LDS
        double *Rp = lds + C::oR;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int sj = j >> 4, lj = j & 15;
            double acc[SM];
#pragma unroll
            for (int s = 0; s < SM; ++s) acc[s] = 0.0;
#pragma unroll
            for (int c = 0; c < N; ++c) {
                const double gb = rb(st.Gh[sj][c], lj);
#pragma unroll
                for (int s = 0; s < SM; ++s)
                    if (16 * s + 15 >= j) acc[s] = fma(st.Gh[s][c], gb, acc[s]);
            }
            if (E > 0) {
#pragma unroll
                for (int e = 0; e < C::EC; ++e) {
                    const double wb = rb(W[sj][e], lj);
#pragma unroll
                    for (int s = 0; s < SM; ++s)
                        if (16 * s + 15 >= j) acc[s] = fma(-W[s][e], wb, acc[s]);
                }
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
    STAMP(P, 5);
}

// Right-hand side of the z-part of solve_kkt in hat coordinates (see dqp_pdipm.hip kkt_wz):
//   g = rz - rs/d - Gh (rxh - At^T (At rxh - ryt)).   Touches Gh/At only (not T).
template <class C>
__device__ __forceinline__ void kkt_rhs(const State<C> &st, const double (&rxh)[C::SN],
                                        const double (&rsd)[C::SM], const double (&rz)[C::SM],
                                        const double (&ryt)[C::SE], double (&g)[C::SM], int r)
{
    double u[C::SN];
#pragma unroll
    for (int s = 0; s < C::SN; ++s) u[s] = rxh[s];
    if (C::E > 0) {
        double t[C::SE], at[C::SN];
        mv_nat<C::SE, C::N, C::SN>(st.Ah, rxh, t, false);
#pragma unroll
        for (int s = 0; s < C::SE; ++s) t[s] -= ryt[s];
        mv_tr<C::SE, C::N, C::SN>(st.Ah, t, at, r);
#pragma unroll
        for (int s = 0; s < C::SN; ++s) u[s] -= at[s];
    }
    double gu[C::SM];
    mv_nat<C::SM, C::N, C::SN>(st.Gh, u, gu, false);
#pragma unroll
    for (int s = 0; s < C::SM; ++s) g[s] = rz[s] - rsd[s] - gu[s];
}

// x/y-part: dxh = -q + At^T (At q - ryt), dyt = -(At q - ryt), q = rxh + Gh^T wz
// Also returns gtw = Gh^T wz and atd = At^T dyt, which the caller uses to keep Gh^T z and
// At^T yt up to date without recomputing them every iteration.
template <class C>
__device__ __forceinline__ void kkt_xy(const State<C> &st, const double (&rxh)[C::SN],
                                       const double (&ryt)[C::SE], const double (&wz)[C::SM],
                                       double (&dxh)[C::SN], double (&dyt)[C::SE],
                                       double (&gtw)[C::SN], double (&atd)[C::SN], int r)
{
    double q[C::SN];
    mv_tr<C::SM, C::N, C::SN>(st.Gh, wz, gtw, r);
#pragma unroll
    for (int s = 0; s < C::SN; ++s) { q[s] = gtw[s] + rxh[s]; dxh[s] = -q[s]; atd[s] = 0.0; }
#pragma unroll
    for (int s = 0; s < C::SE; ++s) dyt[s] = 0.0;
    if (C::E > 0) {
        double e[C::SE], at[C::SN];
        mv_nat<C::SE, C::N, C::SN>(st.Ah, q, e, false);
#pragma unroll
        for (int s = 0; s < C::SE; ++s) { e[s] -= ryt[s]; dyt[s] = -e[s]; }
        mv_tr<C::SE, C::N, C::SN>(st.Ah, e, at, r);
#pragma unroll
        for (int s = 0; s < C::SN; ++s) { dxh[s] += at[s]; atd[s] = -at[s]; }
    }
}

template <class C>
__global__ __launch_bounds__(64) void forward_kernel(KParams P)
{
    constexpr int N = C::N, M = C::M, E = C::E, SN = C::SN, SM = C::SM, SE = C::SE;
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

    term_zero_acc(P);
    State<C> st;
    setup<C>(P, qp, r, lds, st);

    bool inN[SN], inM[SM], inE[SE];
#pragma unroll
    for (int s = 0; s < SN; ++s) inN[s] = r + 16 * s < N;
#pragma unroll
    for (int s = 0; s < SM; ++s) inM[s] = r + 16 * s < M;
#pragma unroll
    for (int s = 0; s < SE; ++s) inE[s] = r + 16 * s < E;

    double ph[SN], hh[SM], bt[SE];
#pragma unroll
    for (int s = 0; s < SN; ++s) ph[s] = inN[s] ? P.p[qp * P.sp + r + 16 * s] : 0.0;
#pragma unroll
    for (int s = 0; s < SM; ++s) hh[s] = inM[s] ? P.h[qp * P.sh + r + 16 * s] : 0.0;
#pragma unroll
    for (int s = 0; s < SE; ++s) bt[s] = (E > 0 && inE[s]) ? P.b[qp * P.sb + r + 16 * s] : 0.0;
    tri_solve<SN, N>(lds + C::oLq, st.rdq, ph, r);
    if (E > 0) tri_solve<SE, C::EC>(lds + C::oL1, st.rd1, bt, r);
    double *dummy = lds + C::oDummy + r;
    vec_put<SN>(lds + C::oPh, ph, N, r, dummy);
    vec_put<SM>(lds + C::oH, hh, M, r, dummy);
    if (E > 0) vec_put<SE>(lds + C::oBt, bt, E, r, dummy);

    double T[SM][M], rdu[SM];
    double xh[SN], s_[SM], z[SM], yt[SE];
    {   // initial point: d = 1, solve_kkt(p, 0, -h, -b)                    batch.py:60-74
        double one[SM], zero[SM], mh[SM], mb[SE], wz[SM];
#pragma unroll
        for (int s = 0; s < SM; ++s) { one[s] = inM[s] ? 1.0 : 0.0; zero[s] = 0.0; mh[s] = -hh[s]; }
#pragma unroll
        for (int s = 0; s < SE; ++s) mb[s] = -bt[s];
        kkt_rhs<C>(st, ph, zero, mh, mb, wz, r);
        __builtin_amdgcn_sched_barrier(0);
        factor_T<C>(lds, T, st.rdiag, one, rdu, r);
        lu_solve<SM, M>(T, rdu, wz, r);
        __builtin_amdgcn_sched_barrier(0);
        double gt0[SN], at0[SN];
        kkt_xy<C>(st, ph, mb, wz, xh, yt, gt0, at0, r);
        double ms = INFINITY, mz = INFINITY;
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            z[s] = inM[s] ? wz[s] : 0.0;
            s_[s] = inM[s] ? -wz[s] : 0.0;
            ms = fmin(ms, inM[s] ? s_[s] : INFINITY);
            mz = fmin(mz, inM[s] ? z[s] : INFINITY);
        }
        ms = row_min(ms); mz = row_min(mz);                                 // batch.py:76-86
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            if (ms < 0.0 && inM[s]) s_[s] -= ms - 1.0;
            if (mz < 0.0 && inM[s]) z[s] -= mz - 1.0;
        }
    }

    STAMP(P, 6);
    // gz = Gh^T z and ay = At^T yt are carried incrementally (updated with the step's own
    // transposed products), so each iteration needs three transposed mat-vecs instead of five.
    double gz[SN], ay[SN];
    mv_tr<SM, N, SN>(st.Gh, z, gz, r);
#pragma unroll
    for (int s = 0; s < SN; ++s) ay[s] = 0.0;
    if (E > 0) mv_tr<SE, N, SN>(st.Ah, yt, ay, r);
    double best = INFINITY;
    vec_put<SN>(lds + C::oBx, xh, N, r, dummy);
    vec_put<SM>(lds + C::oBs, s_, M, r, dummy);
    vec_put<SM>(lds + C::oBz, z, M, r, dummy);
    if (E > 0) vec_put<SE>(lds + C::oBy, yt, E, r, dummy);
    bool have_best = false, done = false;
    int nNot = 0, iters = 0;

    for (int it = 0; it < maxIter; ++it) {
        // residuals in hat coordinates                                    batch.py:93-108
        double rxh[SN], ryt[SE], rz[SM], tmpN[SN];
        vec_get<SN>(lds + C::oPh, tmpN, N, r);
#pragma unroll
        for (int s = 0; s < SN; ++s) rxh[s] = xh[s] + tmpN[s] + gz[s] + ay[s];
#pragma unroll
        for (int s = 0; s < SE; ++s) ryt[s] = 0.0;
        if (E > 0) {
            mv_nat<SE, N, SN>(st.Ah, xh, ryt, false);
            double btl[SE];
            vec_get<SE>(lds + C::oBt, btl, E, r);
#pragma unroll
            for (int s = 0; s < SE; ++s) ryt[s] -= btl[s];
        }
        mv_nat<SM, N, SN>(st.Gh, xh, rz, false);
        double sz = 0.0, nz2 = 0.0, nx2 = 0.0, ny2 = 0.0;
        double hl[SM];
        vec_get<SM>(lds + C::oH, hl, M, r);
#pragma unroll
        for (int s = 0; s < SM; ++s) {
            rz[s] = inM[s] ? rz[s] + s_[s] - hl[s] : 0.0;
            sz = fma(s_[s], z[s], sz);
            nz2 = fma(rz[s], rz[s], nz2);
        }
        {
            double rx[SN];
            tri_mv<SN, N>(lds + C::oLq, rxh, rx, r);                        // rx = Lq rxh
#pragma unroll
            for (int s = 0; s < SN; ++s) nx2 = fma(rx[s], rx[s], nx2);
            if (E > 0) {
                double ry[SE];
                tri_mv<SE, C::EC>(lds + C::oL1, ryt, ry, r);
#pragma unroll
                for (int s = 0; s < SE; ++s) ny2 = fma(ry[s], ry[s], ny2);
            }
        }
        sz = row_sum(sz); nz2 = row_sum(nz2); nx2 = row_sum(nx2); ny2 = row_sum(ny2);
        const double mu = fabs(sz * (1.0 / M));
        const double resid = sqrt(nz2) + sqrt(ny2) + sqrt(nx2) + M * mu;
        // best-iterate tracking / per-problem termination (uniform inside a DPP row)
        if (!done) {
            iters = it + 1;
            if (!have_best || resid < best) {
                nNot = 0; have_best = true; best = resid;
                vec_put<SN>(lds + C::oBx, xh, N, r, dummy);
                vec_put<SM>(lds + C::oBs, s_, M, r, dummy);
                vec_put<SM>(lds + C::oBz, z, M, r, dummy);
                if (E > 0) vec_put<SE>(lds + C::oBy, yt, E, r, dummy);
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

        // phase 1 (Gh/At live, T dead): affine right-hand side (rs = z => rs/d = s)
        double dza[SM], dsa[SM];
        kkt_rhs<C>(st, rxh, s_, rz, ryt, dza, r);
        __builtin_amdgcn_sched_barrier(0);
        // phase 2 (T live, Gh/At idle): factor + affine and corrector solves   batch.py:110-181
        // Divisions by s and z go through one reciprocal each per iteration; the step length
        // min_i(-v_i/dv_i | dv_i < 0) (get_step, batch.py:211-214) is 1 / max_i(-dv_i/v_i), so the
        // row reduction runs on products and only the row-uniform result is inverted.  Pad lanes
        // carry zero reciprocals (and exact zeros out of lu_solve), so nothing else is guarded.
        double dinv[SM];
#pragma unroll
        for (int s = 0; s < SM; ++s) dinv[s] = inM[s] ? s_[s] * frcp(z[s]) : 0.0;   // 1/d, d = z/s
        factor_T<C>(lds, T, st.rdiag, dinv, rdu, r);
        if (it == 1) STAMP(P, 10);
        lu_solve<SM, M>(T, rdu, dza, r);
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
        bool zero_step = false;             // DQP_FLAG_STRICT_GET_STEP, see dqp_r16n.hip
        if (strict) {
#pragma unroll
            for (int s = 0; s < SM; ++s) zero_step |= inM[s] && (dza[s] == 0.0 || dsa[s] == 0.0);
        }
        double alpha = frcp(fmax(row_max(tm), 1.0));                           // min(step, 1)
        if (it == 1) STAMP(P, 11);
        double t3 = 0.0;
#pragma unroll
        for (int s = 0; s < SM; ++s)
            t3 = fma(fma(alpha, dsa[s], s_[s]), fma(alpha, dza[s], z[s]), t3);
        t3 = row_sum(t3);
        double sig = t3 / sz;
        sig = sig * sig * sig;
        // corrector: rx = rz = ry = 0, rs = (-mu sig + ds_a dz_a)/s            batch.py:171-181
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
        // phase 3 (Gh/At live, T dead): x / y part of the combined direction
        double dxh[SN], dyt[SE], gtd[SN], atd[SN];
        kkt_xy<C>(st, rxh, ryt, dz, dxh, dyt, gtd, atd, r);
        alpha = frcp(fmax(row_max(tm) * (1.0 / 0.999), 1.0));                  // min(0.999 step, 1)
        if (it == 1) STAMP(P, 13);
        if (it == 0) STAMP(P, 8);
        if (!done) {
#pragma unroll
            for (int s = 0; s < SN; ++s) {
                xh[s] = fma(alpha, dxh[s], xh[s]);
                gz[s] = fma(alpha, gtd[s], gz[s]);
                ay[s] = fma(alpha, atd[s], ay[s]);
            }
#pragma unroll
            for (int s = 0; s < SM; ++s) { s_[s] = fma(alpha, ds[s], s_[s]); z[s] = fma(alpha, dz[s], z[s]); }
#pragma unroll
            for (int s = 0; s < SE; ++s) yt[s] = fma(alpha, dyt[s], yt[s]);
        }
    }

    if (P.hist && r == 0 && live) hist_fill(P, qp, iters);
    // back to the caller's coordinates: x = Lq^-T xh, y = L1^-T yt
    STAMP(P, 7);
    double bxh[SN], bs[SM], bz[SM], byt[SE];
    vec_get<SN>(lds + C::oBx, bxh, N, r);
    vec_get<SM>(lds + C::oBs, bs, M, r);
    vec_get<SM>(lds + C::oBz, bz, M, r);
#pragma unroll
    for (int s = 0; s < SE; ++s) byt[s] = 0.0;
    if (E > 0) vec_get<SE>(lds + C::oBy, byt, E, r);
    tri_solve_T<SN, N>(lds + C::oLq, st.rdq, bxh, r);
    if (E > 0) tri_solve_T<SE, C::EC>(lds + C::oL1, st.rd1, byt, r);
    if (live) {
#pragma unroll
        for (int s = 0; s < SN; ++s)
            if (inN[s]) P.zhat[qp * N + r + 16 * s] = bxh[s];
#pragma unroll
        for (int s = 0; s < SM; ++s)
            if (inM[s]) { P.lam[qp * M + r + 16 * s] = bz[s]; P.slack[qp * M + r + 16 * s] = bs[s]; }
        if (E > 0) {
#pragma unroll
            for (int s = 0; s < SE; ++s)
                if (inE[s]) P.nu[qp * E + r + 16 * s] = byt[s];
        }
        if (r == 0) {
            if (P.info) { P.info[2 * qp] = st.status; P.info[2 * qp + 1] = iters; }
            if (P.best_resid) P.best_resid[qp] = best;
        }
    }
    STAMP(P, 14);
}

template <class C>
__global__ __launch_bounds__(64) void backward_kernel(KParams P)
{
    constexpr int N = C::N, M = C::M, E = C::E, SN = C::SN, SM = C::SM, SE = C::SE;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x, r = lane & 15, qrow = lane >> 4;
    long long qp = (long long)blockIdx.x * 4 + qrow;
    const bool live = qp < P.B;
    if (!live) qp = P.B - 1;
    double *lds = sm + qrow * C::ldsQPpad;

    State<C> st;
    setup<C>(P, qp, r, lds, st);

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

    // solve_kkt(rx = dl_dzhat, 0, 0, 0)
    tri_solve<SN, N>(lds + C::oLq, st.rdq, g, r);                       // rxh = Lq^-1 g
    double zeroM[SM], zeroE[SE], dlam[SM], dxh[SN], dyt[SE];
#pragma unroll
    for (int s = 0; s < SM; ++s) zeroM[s] = 0.0;
#pragma unroll
    for (int s = 0; s < SE; ++s) zeroE[s] = 0.0;
    kkt_rhs<C>(st, g, zeroM, zeroM, zeroE, dlam, r);
    __builtin_amdgcn_sched_barrier(0);
    {
        double T[SM][M], rdu[SM];
        factor_T<C>(lds, T, st.rdiag, dinv, rdu, r);
        lu_solve<SM, M>(T, rdu, dlam, r);
    }
    __builtin_amdgcn_sched_barrier(0);
    double gtd[SN], atd[SN];
    kkt_xy<C>(st, g, zeroE, dlam, dxh, dyt, gtd, atd, r);
    tri_solve_T<SN, N>(lds + C::oLq, st.rdq, dxh, r);                   // dx = Lq^-T dxh
    if (E > 0) tri_solve_T<SE, C::EC>(lds + C::oL1, st.rd1, dyt, r);    // dnu

    if (!live) return;
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
            if (P.db && inE[s]) P.db[qp * E + r + 16 * s] = -dyt[s];
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
            const double dni = BC(dyt, i), ni = BC(nu, i);
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

}  // namespace r16

// One translation unit per size: _build.py compiles this file once for every entry of
// R16_SIZES with -DDQP_R16_N/M/E (in parallel) and dqp_dispatch.hip routes to the matching
// dqp::r16_forward_<N>_<M>_<E>.
#if !defined(DQP_R16_N) || !defined(DQP_R16_M) || !defined(DQP_R16_E)
#error "compile with -DDQP_R16_N=.. -DDQP_R16_M=.. -DDQP_R16_E=.."
#endif
#define DQP_CAT2(a, n, m, e) a##n##_##m##_##e
#define DQP_CAT(a, n, m, e) DQP_CAT2(a, n, m, e)

// -DDQP_R16_BWD selects the backward kernel: forward and backward of one size are separate
// objects so that the two long compiles run side by side.
#ifndef DQP_R16_BWD
int DQP_CAT(r16_forward_, DQP_R16_N, DQP_R16_M, DQP_R16_E)(const KParams &P, void *stream)
{
    using C = r16::Cfg<DQP_R16_N, DQP_R16_M, DQP_R16_E>;
    return r16::launch<C>(r16::forward_kernel<C>, P, stream);
}
#else
int DQP_CAT(r16_backward_, DQP_R16_N, DQP_R16_M, DQP_R16_E)(const KParams &P, void *stream)
{
    using C = r16::Cfg<DQP_R16_N, DQP_R16_M, DQP_R16_E>;
    return r16::launch<C>(r16::backward_kernel<C>, P, stream);
}
#endif

}  // namespace dqp
