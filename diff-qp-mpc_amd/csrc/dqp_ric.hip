// dqp_ric.hip -- stage-wise PDIPM for MPC-structured QPs of any horizon: the same Mehrotra
// predictor-corrector iteration as the dense kernels (reference: qpth/solvers/pdipm/batch.py:46-208,
// batch_LU.py:29-201 behind qpth/qp_wrapper.py:299-335), with every KKT solve done by a Riccati
// recursion over the knots instead of a dense factorisation -- O(T (n+m)^3) instead of
// O(T^3 (n+m)^3), which is what makes BASELINE config 4 (n = 12, m = 4, T = 30: nz = 480) a
// one-kernel problem (SURVEY.md §8 f1 / row g).
//
// Problem (qp_wrapper.py:638-679, never assembled): knots tau_t = [x_t ; u_t], t = 0..T-1,
//     min  sum_t 1/2 tau_t' C_t tau_t + c_t' tau_t
//     s.t. F_t tau_t - x_{t+1} = -f_t  (t < T-1),   x_0 = x0,   u_lower <= u_t <= u_upper
// with the reference's orderings: z = [tau_0 .. tau_{T-1}], y = [dynamics rows t = 0..T-2 ; x_0 rows],
// lam = [upper rows, t-major ; lower rows, t-major].
//
// One KKT solve (batch.py:351-374 solve_kkt, any right-hand side (rx, rs, rz, ry)):
//     eliminate ds, dz:   Phi_t = C_t + diag(0, d_up + d_lo),   rhs1 = -rx + G'(rs - D rz)
//     [Phi A' ; A 0] [dx ; dy] = [rhs1 ; -ry]   is an LQR problem, solved by
//       backward  H_t = Phi_t + F_t' P_{t+1} F_t ;  partial Cholesky of H_t on its control pivots leaves
//                 P_t (state block), Lxu, Luu in place ;  the same elimination on the vector
//                 h_t = q_t + F_t'(P_{t+1} e_t + p_{t+1}) leaves p_t and Luu^-1 h_u
//       forward   du_t = -Luu^-T (Luu^-1 h_u + Lxu' dx_t) ;  dx_{t+1} = F_t dtau_t + e_t ;
//                 dy_t = P_{t+1} dx_{t+1} + p_{t+1} ;  dy_init = -(P_0 dx_0 + p_0)
//     (q = -rhs1, e = ry, dx_0 = -ry_init), then ds = -rz - G dx, dz = -rs - D ds.
// The factorisation is shared by the affine and the corrector solve of an iteration.
//
// Layout: one QP per 16-lane DPP row (n + m <= 16: a knot is one distributed vector, a stage matrix
// one row per lane), four QPs per wavefront, iterates / directions / per-knot factors streamed through
// a caller workspace (dqp_mpc_qp_workspace_bytes) -- the kernel is HBM-stream bound by C, F and the
// factors (~60 k doubles per QP and iteration at config 4), so every knot's matrices are prefetched one
// knot ahead by LDS-DMA (Stage<C>); short horizons run entirely out of LDS (Cfg<n, m, true>).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/dqp.h"
#include "dqp_common.h"
#include "dqp_r16_prims.h"
#include "dqp_dyn_models.h"

namespace dqp {
namespace ric {

using namespace dqp::r16;

// WSL: the per-QP workspace (iterates, directions, factors) lives in LDS instead of the caller's buffer --
// the forward kernel of short horizons, where it fits beside the stage and a knot step is latency, not bandwidth
template <int NX_, int NU_, bool WSL_ = false> struct Cfg {
    static constexpr int NX = NX_, NU = NU_, NT = NX_ + NU_;
    static constexpr bool WSL = WSL_;
    static_assert(NT <= 16, "a knot must fit a 16-lane DPP row");
};

// per-QP workspace (doubles); everything knot-major
struct Lay {
    int X, Y, SU, SL, ZU, ZL;            // iterate
    int RY, RZU, RZL;                    // residuals of the iterate the affine forward sweep reads (rx is consumed where it is formed)
    int DX, DY, DSU, DSL, DZU, DZL;      // step (affine, then affine + corrector)
    int BX, BY, BSU, BSL, BZU, BZL;      // best iterate
    int FACP, FACL, PV, YB;              // per knot: P_t (NX x NX), [Lxu ; Luu] (NT x NU, 1 / L_jj on the diagonal), p_t (NX), Luu^-1 h_u (NU)
    int total;
};
__host__ __device__ inline Lay layout(int nx, int nu, int T)
{
    const int nt = nx + nu;
    Lay L;
    int o = 0;
    auto take = [&](int n) { const int at = o; o += n; return at; };
    L.X = take(T * nt); L.Y = take(T * nx); L.SU = take(T * nu); L.SL = take(T * nu); L.ZU = take(T * nu); L.ZL = take(T * nu);
    L.RY = take(T * nx); L.RZU = take(T * nu); L.RZL = take(T * nu);
    L.DX = take(T * nt); L.DY = take(T * nx); L.DSU = take(T * nu); L.DSL = take(T * nu); L.DZU = take(T * nu); L.DZL = take(T * nu);
    L.BX = take(T * nt); L.BY = take(T * nx); L.BSU = take(T * nu); L.BSL = take(T * nu); L.BZU = take(T * nu); L.BZL = take(T * nu);
    o = (o + 1) & ~1; L.FACP = take(T * nx * nx); o = (o + 1) & ~1; L.FACL = take(T * nt * nu);     // 16-byte aligned: DMA sources
    L.PV = take(T * nx); L.YB = take(T * nu);
    L.total = (o + 1) & ~1;
    return L;
}

enum Mode { INIT = 0, AFFINE = 1, CORRECTOR = 2, ADJOINT = 3 };

extern __shared__ __attribute__((aligned(16))) double lds_dyn[];       // Cfg<., ., true>: [C | F | c | f | four workspaces]

template <class C> struct Ctx {
    const KParams &P;
    double *w;                      // this QP's workspace
    Lay L;
    long long qp;
    int r, T;
    bool xl, ul, live;              // lane holds a state row / a control row / a real problem
    int a;                          // control index of a control lane (0 otherwise)
    double uu, ulo;                 // bounds of this lane's control
    int lane, g, qmax;              // lane, place of its problem in the wavefront, last place with a problem of its own
    long long qp0;                  // problem at place 0
    double *w0;                     // workspace of place 0 (places are L.total doubles apart)
    double *img;                    // the wavefront's LDS stage (Stage<C>); WSL: every knot's C_t, then every F_t
    int rf;                         // WSL: where the F_t images start in img
    const double *lc, *lf;          // WSL: c and f of the four problems, [(t * 4 + place) * NT or NX + r]
    __device__ const double *imgC(int t) const { return C::WSL ? img + t * (4 * C::NT * C::NT) : img; }
    __device__ const double *imgF(int t) const;
    __device__ double cvec(int t) const { return C::WSL ? lc[(t * 4 + g) * C::NT + r] : P.mc[((long long)t * P.B + qp) * C::NT + r]; }
    __device__ double fvec(int t) const { return C::WSL ? lf[(t * 4 + g) * C::NX + r] : P.mf[((long long)t * P.B + qp) * C::NX + r]; }
    // knot t of the wavefront's four problems: contiguous in the (T, B, ., .) inputs
    __device__ const double *Cblk(int t) const { return P.mC + ((long long)t * P.B + qp0) * (C::NT * C::NT); }
    __device__ const double *Fblk(int t) const { return P.mF + ((long long)t * P.B + qp0) * (C::NX * C::NT); }
    __device__ const double *Pblk(int t) const { return w0 + L.FACP + (long long)t * (C::NX * C::NX); }
    __device__ const double *Lblk(int t) const { return w0 + L.FACL + (long long)t * (C::NT * C::NU); }
};

// A copy of the context whose per-lane address roots the compiler must treat as new values: every
// sweep is inlined into the iteration loop, and without this LICM hoists the address arithmetic of all
// of them (array bases, DMA source offsets, swizzled LDS addresses: ~150 registers) out of that loop and
// spills it; the reloads then sit in every knot.
template <class C> __device__ __forceinline__ Ctx<C> fresh(const Ctx<C> &K0)
{
    Ctx<C> K = K0;
    if constexpr (C::WSL) {         // an LDS pointer stays an LDS pointer (ds_ instead of flat_ accesses)
        int off = (int)(K0.w - lds_dyn);
        asm volatile("" : "+v"(K.r), "+v"(K.lane), "+v"(K.g), "+v"(off));
        K.w = lds_dyn + off;
    } else {
        asm volatile("" : "+v"(K.r), "+v"(K.lane), "+v"(K.g), "+v"(K.w));
    }
    K.xl = K.r < C::NX;
    K.ul = K.r >= C::NX && K.r < C::NT;
    K.a = K.ul ? K.r - C::NX : 0;
    return K;
}

// ---------------------------------------------------------------------------------------------
// LDS stage.  A sweep visits the knots one after the other and every knot is a chain
// "load its matrices -> a few hundred FMAs -> store": with the loads issued at the knot itself a
// wavefront waits out several memory round trips per knot (12 us per knot at config 4, measured).
// The matrices of knot t -+ 1 are therefore fetched while knot t computes, by LDS-DMA
// (global_load_lds: no destination registers, the kernel has none to spare) into one image per
// wavefront; the per-knot vectors are prefetched through registers.  An image holds the knot's
// NT x NT matrix (C_t, or the factor rows of the knot) and its NX x NT matrix F_t for the four problems.
// The DMA writes LDS lane-linearly (wave base + lane * PIECE), so the XOR swizzle that keeps the
// 16-lane row reads off a single bank group is applied to the SOURCE address of each piece.
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// one kind of matrix in the stage: rows of RL doubles
template <int RL> struct Img {
    static constexpr int PIECE = (RL % 2 == 0) ? 16 : 4;            // bytes per lane of one DMA instruction
    static constexpr int PPR = RL * 8 / PIECE;                       // pieces per matrix row
    // rows of 2^k 16-byte pieces start on few distinct bank groups: piece j of row `row` sits at j ^ swz(row)
    static constexpr int SWZ_DIV = (PIECE == 16 && PPR > 1 && PPR <= 8 && (PPR & (PPR - 1)) == 0) ? 16 / PPR : 0;
    static constexpr int doubles(int rows) { return (4 * rows * PPR + 63) / 64 * 64 * PIECE / 8; }   // image of four problems
    __device__ __forceinline__ static int swz(int row) { return SWZ_DIV ? (row / (SWZ_DIV ? SWZ_DIV : 1)) & (PPR - 1) : 0; }
    // columns [C0, C1) of row `row` of place g.  With the swizzle, piece j of the row is at index
    // (base | s << 1) ^ (j << 1) (base is a multiple of the row length 2 PPR, a power of two there): one
    // XOR per piece off a single register, which is handed to the compiler as a new value at every call --
    // kept as loop invariants the piece addresses of a kernel's sweeps are ~60 registers, spilled and
    // reloaded one by one in front of the reads.
    template <int ROWS, int C0, int C1, int N>
    __device__ __forceinline__ static void cols(const double *img, int g, int row, double (&out)[N])
    {
        static_assert(N == C1 - C0, "");
        if constexpr (PIECE == 16) {
            int root = ((g * ROWS + row) * RL) | (swz(row) << 1);
            asm volatile("" : "+v"(root));
#pragma unroll
            for (int j = C0 / 2; j <= (C1 - 1) / 2; ++j) {
                const double2 v = *reinterpret_cast<const double2 *>(img + (SWZ_DIV ? (root ^ (j << 1)) : root + (j << 1)));
                if (2 * j >= C0 && 2 * j < C1) out[2 * j - C0] = v.x;
                if (2 * j + 1 >= C0 && 2 * j + 1 < C1) out[2 * j + 1 - C0] = v.y;
            }
        } else {
            int root = (g * ROWS + row) * RL;
            asm volatile("" : "+v"(root));
#pragma unroll
            for (int c = C0; c < C1; ++c) out[c - C0] = img[root + c];
        }
    }
    // column `col` (rows 0 .. ROWS-1) of place g
    template <int ROWS>
    __device__ __forceinline__ static void column(const double *img, int g, int col, double (&out)[ROWS])
    {
        int root = g * ROWS * RL + col;
        asm volatile("" : "+v"(root));
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            if constexpr (PIECE == 16 && SWZ_DIV != 0) out[i] = img[(root ^ (((i / SWZ_DIV) & (PPR - 1)) << 1)) + i * RL];
            else out[i] = img[root + i * RL];
        }
    }
    // issue the DMA of one knot's ROWS x RL matrices: place q's matrix is at base + min(q, qmax) * qstride_bytes
    template <int ROWS>
    __device__ __forceinline__ static void fetch(double *img, const double *base, int qstride_bytes, int qmax, int lane)
    {
        constexpr int PIECES = 4 * ROWS * PPR, KN = (PIECES + 63) / 64;
#pragma unroll
        for (int k = 0; k < KN; ++k) {
            int p = k * 64 + lane;
            if constexpr (PIECES % 64 != 0) p = p < PIECES ? p : PIECES - 1;     // tail lanes refill the padding
            const int rowg = p / PPR, j = p % PPR, q = rowg / ROWS, row = rowg % ROWS;
            const int off = (q < qmax ? q : qmax) * qstride_bytes + (row * PPR + (j ^ swz(row))) * PIECE;
            gvoid_t *src = (gvoid_t *)(reinterpret_cast<const char *>(base) + off);
            lvoid_t *dst = (lvoid_t *)(img + k * (64 * PIECE / 8));
            if constexpr (PIECE == 16) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
            else __builtin_amdgcn_global_load_lds(src, dst, 4, 0, 0);
        }
    }
    // the same for `count` consecutive knots at once (knot stride tstride_bytes in the source), images back to
    // back without padding: the resident copy of a short-horizon problem
    template <int ROWS>
    __device__ __forceinline__ static void fetch_all(double *img, const double *base, long long tstride_bytes, int qstride_bytes,
                                                     int qmax, int lane, int count)
    {
        constexpr int PIECES = 4 * ROWS * PPR;
        const int total = count * PIECES;
        for (int k = 0; k * 64 < total; ++k) {
            int p = k * 64 + lane;
            p = p < total ? p : total - 1;
            const int t = p / PIECES, pp = p % PIECES;
            const int rowg = pp / PPR, j = pp % PPR, q = rowg / ROWS, row = rowg % ROWS;
            const long long off = t * tstride_bytes + (q < qmax ? q : qmax) * qstride_bytes + (row * PPR + (j ^ swz(row))) * PIECE;
            gvoid_t *src = (gvoid_t *)(reinterpret_cast<const char *>(base) + off);
            lvoid_t *dst = (lvoid_t *)(img + k * (64 * PIECE / 8));
            if constexpr (PIECE == 16) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
            else __builtin_amdgcn_global_load_lds(src, dst, 4, 0, 0);
        }
    }
    static constexpr __host__ __device__ int resident_doubles(int rows, int count) { return (count * 4 * rows * PPR + 63) / 64 * 64 * PIECE / 8; }
};

// the wavefront's image: [C_t | F_t] while factorising, [P_t | L_t | F_t] in the vector sweeps
template <class C> struct Stage {
    static constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    using MC = Img<NT>;             // C_t: NT rows of NT
    using MF = Img<NT>;             // F_t: NX rows of NT
    using MP = Img<NX>;             // cost-to-go P_t: NX rows of NX
    using ML = Img<NU>;             // [Lxu ; Luu] with 1 / L_jj on the diagonal: NT rows of NU
    static constexpr int OL = MP::doubles(NX);
    static constexpr int OF = MC::doubles(NT) > OL + ML::doubles(NT) ? MC::doubles(NT) : OL + ML::doubles(NT);
    static constexpr int TOTAL = OF + MF::doubles(NX);
    // WSL (everything of a short-horizon problem in LDS): [C_0 .. C_{T-1} | F_0 .. F_{T-2} | c | f | four workspaces]
    static __host__ __device__ int res_f(int T) { return MC::resident_doubles(NT, T); }
    static __host__ __device__ int res_c(int T) { return res_f(T) + MF::resident_doubles(NX, T - 1 > 0 ? T - 1 : 1); }
    static __host__ __device__ int res_fv(int T) { return res_c(T) + T * 4 * NT; }
    static __host__ __device__ int res_ws(int T) { return (res_fv(T) + T * 4 * NX + 1) & ~1; }
};

template <class C> __device__ __forceinline__ const double *Ctx<C>::imgF(int t) const
{
    return C::WSL ? img + rf + t * (4 * C::NX * C::NT) : img + Stage<C>::OF;
}

// y[r] = sum_c row[c] * v[c]  (row = this lane's matrix row, v distributed)
template <int N> __device__ __forceinline__ double mv_row(const double (&row)[N], double v)
{
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < N; ++c) acc = fma(row[c], rb(v, c), acc);
    return acc;
}

// H += F' (P F) for this lane's row of H: Pn = this lane's row of P_{t+1} (state lanes, zero elsewhere),
// frow = its row of F_t (state lanes), fcol = its column of F_t
template <class C>
__device__ __forceinline__ void add_FtPF(double (&H)[C::NT], const double (&Pn)[C::NX], const double (&frow)[C::NT],
                                         const double (&fcol)[C::NX])
{
    constexpr int NX = C::NX, NT = C::NT;
    double PF[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {                  // PF = P_{t+1} F_t, row-distributed over the state lanes
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < NX; ++c) acc = fma(Pn[c], rb(frow[j], c), acc);
        PF[j] = acc;
    }
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        double acc = H[b];
#pragma unroll
        for (int i = 0; i < NX; ++i) acc = fma(fcol[i], rb(PF[b], i), acc);
        H[b] = acc;
    }
}

// this lane's row of C_t, its row and its column of F_t, out of the stage
template <class C>
__device__ __forceinline__ void stage_rows(const Ctx<C> &K, int t, bool withF, double (&H)[C::NT], double (&frow)[C::NT],
                                           double (&fcol)[C::NX])
{
    using S = Stage<C>;
    constexpr int NX = C::NX, NT = C::NT;
    const int r = K.r;
    S::MC::template cols<NT, 0, NT>(K.imgC(t), K.g, r < NT ? r : 0, H);
    if (withF) {
        S::MF::template cols<NX, 0, NT>(K.imgF(t), K.g, K.xl ? r : 0, frow);
        S::MF::template column<NX>(K.imgF(t), K.g, r < NT ? r : 0, fcol);
#pragma unroll
        for (int c = 0; c < NT; ++c) frow[c] = K.xl ? frow[c] : 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) fcol[i] = r < NT ? fcol[i] : 0.0;
    }
}
template <class C> __device__ __forceinline__ void fetch_CF(const Ctx<C> &K, int t)
{
    using S = Stage<C>;
    if constexpr (C::WSL) return;
    S::MC::template fetch<C::NT>(K.img, K.Cblk(t), C::NT * C::NT * 8, K.qmax, K.lane);
    if (t < K.T - 1) S::MF::template fetch<C::NX>(K.img + S::OF, K.Fblk(t), C::NX * C::NT * 8, K.qmax, K.lane);
}
// the factor rows of knot t (P_t only where the sweep multiplies by it) and F_t
template <class C, bool WITHP> __device__ __forceinline__ void fetch_facF(const Ctx<C> &K, int t)
{
    using S = Stage<C>;
    if constexpr (C::WSL) return;
    if (WITHP) S::MP::template fetch<C::NX>(K.img, K.Pblk(t), K.L.total * 8, 3, K.lane);
    S::ML::template fetch<C::NT>(K.img + S::OL, K.Lblk(t), K.L.total * 8, 3, K.lane);
    if (t < K.T - 1) S::MF::template fetch<C::NX>(K.img + S::OF, K.Fblk(t), C::NX * C::NT * 8, K.qmax, K.lane);
}
// this lane's row of [Lxu ; Luu] of knot t and, on a control lane, 1 / L_jj (kept on the diagonal)
template <class C> __device__ __forceinline__ double stage_L(const Ctx<C> &K, int t, double (&lrow)[C::NU])
{
    using S = Stage<C>;
    if constexpr (C::WSL) {
        const double *o = K.w + K.L.FACL + (t * C::NT + (K.r < C::NT ? K.r : 0)) * C::NU;
#pragma unroll
        for (int b = 0; b < C::NU; ++b) lrow[b] = o[b];
    } else {
        S::ML::template cols<C::NT, 0, C::NU>(K.img + S::OL, K.g, K.r < C::NT ? K.r : 0, lrow);
    }
    double rd = 0.0;
#pragma unroll
    for (int b = 0; b < C::NU; ++b) rd = (K.ul && K.a == b) ? lrow[b] : rd;
    return rd;
}
// this lane's row of P_t (state lanes)
template <class C> __device__ __forceinline__ void stage_P(const Ctx<C> &K, int t, double (&prow)[C::NX])
{
    using S = Stage<C>;
    if constexpr (C::WSL) {
        const double *o = K.w + K.L.FACP + (t * C::NX + (K.xl ? K.r : 0)) * C::NX;
#pragma unroll
        for (int c = 0; c < C::NX; ++c) prow[c] = o[c];
    } else {
        S::MP::template cols<C::NX, 0, C::NX>(K.img, K.g, K.xl ? K.r : 0, prow);
    }
}
// the factor rows of one knot out of the registers: H = [P | Lxu] on the state lanes, [. | Luu] on the control lanes
template <class C> __device__ __forceinline__ void store_fac(const Ctx<C> &K, double *w, int t, const double (&H)[C::NT], double rdj)
{
    constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    const int r = K.r;
    if (K.xl) {
        double *o = w + K.L.FACP + ((long long)t * NX + r) * NX;
#pragma unroll
        for (int c = 0; c < NX; ++c) o[c] = H[c];
    }
    if (r < NT) {
        double *o = w + K.L.FACL + ((long long)t * NT + r) * NU;
#pragma unroll
        for (int b = 0; b < NU; ++b) o[b] = (r == NX + b) ? rdj : H[NX + b];
    }
}

// ---------------------------------------------------------------------------------------------
// backward Riccati sweep on the matrices; d = z/s from the iterate (unit: d = 1, clampd: the
// reference's backward clamps, qp.py:131-134).  Returns false if a control pivot is not positive.
template <class C>
__device__ __forceinline__ bool factor(const Ctx<C> &K0, bool unit, bool clampd)
{
    const Ctx<C> K = fresh<C>(K0);
    constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    const int r = K.r, T = K.T;
    double *w = K.w;
    const Lay &L = K.L;
    double Pn[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) Pn[c] = 0.0;
    bool ok = true;
    auto load_d = [&](int t) {
        double dd = 2.0;
        if (K.ul && !unit) {
            const int iu = t * NU + K.a;
            double su = w[L.SU + iu], sl = w[L.SL + iu], zu = w[L.ZU + iu], zl = w[L.ZL + iu];
            if (clampd) { su = fmax(su, 1e-8); sl = fmax(sl, 1e-8); zu = fmax(zu, 1e-8); zl = fmax(zl, 1e-8); }
            dd = zu * frcp(su) + zl * frcp(sl);
        }
        return dd;
    };
    double dd = load_d(T - 1);
    fetch_CF<C>(K, T - 1);
    for (int t = T - 1; t >= 0; --t) {
        double H[NT], frow[NT], fcol[NX];
        wait_vm();
        stage_rows<C>(K, t, t < T - 1, H, frow, fcol);
        wait_lds();
        double dn = dd;
        if (t > 0) { dn = load_d(t - 1); fetch_CF<C>(K, t - 1); }
#pragma unroll
        for (int c = 0; c < NT; ++c) H[c] = r < NT ? H[c] + ((K.ul && r == c) ? dd : 0.0) : (r == c ? 1.0 : 0.0);
        if (t < T - 1) add_FtPF<C>(H, Pn, frow, fcol);
        // partial Cholesky on the control pivots j = NX .. NT-1
        double rdj_keep = 0.0;
#pragma unroll
        for (int j = NX; j < NT; ++j) {
            const double pj = rb(H[j], j);
            if (!(pj > 0.0)) ok = false;
            const double rdj = frsqrt(pj > 0.0 ? pj : 1.0);
            const double lij = H[j] * rdj;                  // column j of L on every lane (lane j: sqrt(pj))
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                if (c >= NX && c <= j) continue;            // L columns already final
                H[c] = fma(-lij, rb(lij, c), H[c]);
            }
            H[j] = lij;
            if (r == j) rdj_keep = rdj;
        }
        store_fac<C>(K, w, t, H, rdj_keep);
#pragma unroll
        for (int c = 0; c < NX; ++c) Pn[c] = K.xl ? H[c] : 0.0;
        dd = dn;
    }
    return ok;
}

// The true-dynamics residual of every knot (the reference's dyn_res closure, qp_wrapper.py:309,316 ->
// batch_LU.py:97), ry_t = f(x_t, u_t) - x_{t+1}, before the fused sweep: lane r of a problem's row evaluates
// the registered model at knot base + r, sixteen knots of the problem at a time.  (The first version evaluated
// the model inside the fused sweep, every lane of the row the same knot: sixteen times the work, and the
// registers of an RK4 step of the quadrotor spilled around the Riccati update.)
template <class C, class Map> constexpr bool model_fits() { return Map::NX == C::NX && Map::NU == C::NU; }
template <class C, class Map>
__device__ __forceinline__ void model_residuals_of(const Ctx<C> &K)
{
    if constexpr (model_fits<C, Map>()) {
        constexpr int NX = C::NX, NT = C::NT;
        double *w = K.w;
        const Lay &L = K.L;
        for (int base = 0; base < K.T - 1; base += 16) {
            const int t = base + K.r;
            if (t < K.T - 1) {
                double z[NT], xn[NX];
#pragma unroll
                for (int j = 0; j < NT; ++j) z[j] = w[L.X + t * NT + j];
                Map::template step<double>(z, z + NX, K.P.dynDt, xn);
#pragma unroll
                for (int j = 0; j < NX; ++j) w[L.RY + t * NX + j] = xn[j] - w[L.X + (t + 1) * NT + j];
            }
        }
    }
}
template <class C>
__device__ __forceinline__ void model_residuals(const Ctx<C> &K0)
{
    using namespace dqp::dyn;
    const Ctx<C> K = fresh<C>(K0);
    switch (K.P.dynId) {
    case DQP_DYN_PENDULUM1L: model_residuals_of<C, Robot<Pendulum1l>>(K); break;
    case DQP_DYN_CARTPOLE1L: model_residuals_of<C, Robot<Cartpole1l>>(K); break;
    case DQP_DYN_CARTPOLE2L: model_residuals_of<C, Robot<Cartpole2l>>(K); break;
    case DQP_DYN_PENDULUM_EULER: model_residuals_of<C, PendulumEuler>(K); break;
    case DQP_DYN_PENDULUM_DX: model_residuals_of<C, PendulumDx>(K); break;
    case DQP_DYN_REXQUADROTOR: model_residuals_of<C, RexQuadrotor>(K); break;
    default: break;
    }
}

template <class C> constexpr bool has_model()
{
    using namespace dqp::dyn;
    return model_fits<C, Robot<Pendulum1l>>() || model_fits<C, Robot<Cartpole1l>>() || model_fits<C, Robot<Cartpole2l>>() ||
           model_fits<C, PendulumEuler>() || model_fits<C, PendulumDx>() || model_fits<C, RexQuadrotor>();
}

// ---------------------------------------------------------------------------------------------
// One backward sweep for the three things an iteration needs from every knot before it can move:
// the residuals of the iterate (batch.py:93-108: rx = C tau + c + G'z + A'y, rz = G tau + s - h,
// ry = A tau - b), the Riccati factorisation with d = z/s (as factor()) and the affine right-hand side
// pushed through it (as sweep_back<AFFINE>) -- C_t and F_t are read once instead of three times.
template <class C, bool DYN>
__device__ __forceinline__ bool factor_fused(const Ctx<C> &K0, double &nx2, double &nz2, double &ny2, double &sz)
{
    const Ctx<C> K = fresh<C>(K0);
    constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    const int r = K.r, T = K.T;
    double *w = K.w;
    const Lay &L = K.L;
    nx2 = nz2 = ny2 = sz = 0.0;
    double Pn[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) Pn[c] = 0.0;
    double pn = 0.0;
    bool ok = true;
    // the knot's vectors: tau_t, c_t, (s, z)_t, y_{t-1} (t = 0: the multiplier of x_0 = x0), f_t, x0
    enum { V_TAU, V_MC, V_SU, V_SL, V_ZU, V_ZL, V_YP, V_MF, V_X0, V_N };
    auto load_vec = [&](const Ctx<C> &K, int t, double (&v)[V_N]) {
        const int r = K.r;
        double *w = K.w;
#pragma unroll
        for (int i = 0; i < V_N; ++i) v[i] = (i == V_SU || i == V_SL) ? 1.0 : 0.0;
        if (r < NT) { v[V_TAU] = w[L.X + t * NT + r]; v[V_MC] = K.cvec(t); }
        if (K.ul) {
            const int iu = t * NU + K.a;
            v[V_SU] = w[L.SU + iu]; v[V_SL] = w[L.SL + iu]; v[V_ZU] = w[L.ZU + iu]; v[V_ZL] = w[L.ZL + iu];
        }
        if (K.xl) {
            v[V_YP] = w[L.Y + (t >= 1 ? t - 1 : T - 1) * NX + r];
            if (t < T - 1) v[V_MF] = DYN ? w[L.RY + t * NX + r] : K.fvec(t);      // DYN: ry_t itself (model_residuals)
            if (t == 0) v[V_X0] = K.P.mx0[K.qp * NX + r];
        }
    };
    double cur[V_N], nxt[V_N];
    load_vec(K, T - 1, cur);
    fetch_CF<C>(K, T - 1);
    double x_next = 0.0, y_t = 0.0;         // x_{t+1} and y_t on the state lanes, carried from knot t + 1
    for (int t = T - 1; t >= 0; --t) {
        double H[NT], frow[NT], fcol[NX];
        wait_vm();
        stage_rows<C>(K, t, t < T - 1, H, frow, fcol);
        wait_lds();
        if (t > 0) {
            const Ctx<C> Kt = fresh<C>(K);          // the prefetch addresses are recomputed per knot, not kept
            load_vec(Kt, t - 1, nxt);
            fetch_CF<C>(Kt, t - 1);
        }
        double *w;                                    // same for the stores of this knot
        { const Ctx<C> Ks = fresh<C>(K); w = Ks.w; }
        const double tau = cur[V_TAU];
#pragma unroll
        for (int c = 0; c < NT; ++c) H[c] = r < NT ? H[c] : 0.0;
        double rx = mv_row<NT>(H, tau) + cur[V_MC];
        double q = 0.0, e = 0.0;
        if (K.ul) {
            const int iu = t * NU + K.a;
            const double su = cur[V_SU], sl = cur[V_SL], zu = cur[V_ZU], zl = cur[V_ZL];
            rx += zu - zl;
            const double rzu = tau - K.uu + su, rzl = -tau + K.ulo + sl;
            w[L.RZU + iu] = rzu; w[L.RZL + iu] = rzl;
            nz2 = fma(rzu, rzu, fma(rzl, rzl, nz2));
            sz = fma(su, zu, fma(sl, zl, sz));
            const double du_ = zu * frcp(su), dl_ = zl * frcp(sl);
            q = -((zu - du_ * rzu) - (zl - dl_ * rzl));
#pragma unroll
            for (int c = 0; c < NT; ++c) H[c] += (r == c) ? du_ + dl_ : 0.0;
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) H[c] = r < NT ? H[c] : (r == c ? 1.0 : 0.0);
        if (t < T - 1) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) acc = fma(fcol[i], rb(y_t, i), acc);
            rx += acc;
            // ry_t: the linearised dynamics, or the registered model's own residual (computed by model_residuals)
            double fx = 0.0;
            if constexpr (!DYN) fx = mv_row<NT>(frow, tau) + cur[V_MF];       // (row broadcasts: every lane takes part)
            if (K.xl) {
                if constexpr (DYN) e = cur[V_MF];
                else {
                    e = fx - x_next;
                    w[L.RY + t * NX + r] = e;
                }
                ny2 = fma(e, e, ny2);
            }
            add_FtPF<C>(H, Pn, frow, fcol);
        }
        if (K.xl) {
            if (t >= 1) rx -= cur[V_YP];
            else {
                rx += cur[V_YP];
                const double ry = tau - cur[V_X0];
                w[L.RY + (T - 1) * NX + r] = ry;
                ny2 = fma(ry, ry, ny2);
            }
        }
        if (r < NT) nx2 = fma(rx, rx, nx2);           // rx itself is only needed here (q below): not stored
        q += rx;
        // ---- affine right-hand side: h = q + F'(P_{t+1} e + p_{t+1})
        double h = q;
        if (t < T - 1) {
            double v = pn;
#pragma unroll
            for (int c = 0; c < NX; ++c) v = fma(Pn[c], rb(e, c), v);
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) acc = fma(fcol[i], rb(v, i), acc);
            h += acc;
        }
        // ---- partial Cholesky on the control pivots, the same elimination on h
        double rdj_keep = 0.0;
#pragma unroll
        for (int j = NX; j < NT; ++j) {
            const double pj = rb(H[j], j);
            if (!(pj > 0.0)) ok = false;
            const double rdj = frsqrt(pj > 0.0 ? pj : 1.0);
            const double lij = H[j] * rdj;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                if (c >= NX && c <= j) continue;
                H[c] = fma(-lij, rb(lij, c), H[c]);
            }
            H[j] = lij;
            if (r == j) rdj_keep = rdj;
            const double hj = rb(h, j) * rdj;
            h = (r == j) ? hj : ((r < NX || (r > j && r < NT)) ? fma(-lij, hj, h) : h);
        }
        store_fac<C>(K, w, t, H, rdj_keep);
        if (K.xl) w[L.PV + t * NX + r] = h;
        if (K.ul) w[L.YB + t * NU + K.a] = h;
        pn = K.xl ? h : 0.0;
#pragma unroll
        for (int c = 0; c < NX; ++c) Pn[c] = K.xl ? H[c] : 0.0;
        x_next = K.xl ? tau : 0.0;
        y_t = K.xl ? cur[V_YP] : 0.0;
#pragma unroll
        for (int i = 0; i < V_N; ++i) cur[i] = nxt[i];
    }
    return ok;
}

// Right-hand side of one knot for the four uses of the solver, in two steps so that the loads of knot
// t -+ 1 can be issued while knot t computes: rhs_load (memory) and rhs_q / rhs_e (arithmetic).
//   q_t[r] = -rhs1 of the eliminated system, e_t[r] = ry_t
enum { R_A, R_B, R_C, R_D, R_E, R_F, R_G, R_H, R_N };
template <class C, int MODE, bool NEED_Q, bool NEED_E>
__device__ __forceinline__ void rhs_load(const Ctx<C> &K, int t, double (&v)[R_N])
{
    constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    const int r = K.r, T = K.T;
    const double *w = K.w;
    const Lay &L = K.L;
#pragma unroll
    for (int i = 0; i < R_N; ++i) v[i] = (i == R_B || i == R_C) ? 1.0 : 0.0;
    if (MODE == INIT) {            // batch.py:60-74: rx = p, rs = 0, rz = -h, ry = -b with d = 1
        if (NEED_Q && r < NT) v[R_A] = K.cvec(t);
        if (NEED_E && K.xl && t < T - 1) v[R_H] = K.fvec(t);
    } else if (MODE == AFFINE) {   // rx, rs = z, rz, ry of the iterate: q_t is formed and used inside factor_fused
        static_assert(!(MODE == AFFINE && NEED_Q), "the affine backward sweep is part of factor_fused");
        if (NEED_E && K.xl && t < T - 1) v[R_H] = w[L.RY + t * NX + r];
    } else if (MODE == CORRECTOR) {   // rx = 0, rs = (-mu sig + ds_aff dz_aff) / s, rz = ry = 0
        if (NEED_Q && K.ul) {
            const int iu = t * NU + K.a;
            v[R_B] = w[L.SU + iu]; v[R_C] = w[L.SL + iu];
            v[R_D] = w[L.DSU + iu]; v[R_E] = w[L.DZU + iu]; v[R_F] = w[L.DSL + iu]; v[R_G] = w[L.DZL + iu];
        }
        if (NEED_Q && K.xl) v[R_A] = w[L.PV + t * NX + r];       // the affine solve's p_t: see sweep_back
    } else {                       // ADJOINT: rx = dl/dzhat (qp.py:136-141)
        if (NEED_Q && r < NT) v[R_A] = K.P.gin[(K.qp * T + t) * NT + r];            // dl/dtau is (B, T, nt) like tau
    }
}
template <class C, int MODE>
__device__ __forceinline__ double rhs_q(const Ctx<C> &K, const double (&v)[R_N], double musig)
{
    if (MODE == INIT) return v[R_A] - (K.ul ? K.uu + K.ulo : 0.0);
    if (MODE == CORRECTOR) return K.ul ? -((-musig + v[R_D] * v[R_E]) * frcp(v[R_B]) - (-musig + v[R_F] * v[R_G]) * frcp(v[R_C])) : 0.0;
    return v[R_A];
}

// backward vector sweep: p_t and Luu^-1 h_u per stage
template <class C, int MODE>
__device__ __forceinline__ void sweep_back(const Ctx<C> &K0, double musig)
{
    const Ctx<C> K = fresh<C>(K0);
    using S = Stage<C>;
    constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    constexpr bool USE_E = (MODE == INIT || MODE == AFFINE);
    const int r = K.r, T = K.T;
    double *w = K.w;
    const Lay &L = K.L;
    double pn = 0.0;
    double Pn[NX];                  // this lane's row of P_{t+1} (only where e != 0)
#pragma unroll
    for (int c = 0; c < NX; ++c) Pn[c] = 0.0;
    double cur[R_N], nxt[R_N];
    rhs_load<C, MODE, true, USE_E>(K, T - 1, cur);
    fetch_facF<C, USE_E>(K, T - 1);
    for (int t = T - 1; t >= 0; --t) {
        double fcol[NX], lcol[NU], prow[NX];
        wait_vm();
        const double rd = stage_L<C>(K, t, lcol);
        if (USE_E) stage_P<C>(K, t, prow);
        if (t < T - 1) {
            S::MF::template column<NX>(K.imgF(t), K.g, r < NT ? r : 0, fcol);
#pragma unroll
            for (int i = 0; i < NX; ++i) fcol[i] = r < NT ? fcol[i] : 0.0;
        }
        wait_lds();
        if (t > 0) { rhs_load<C, MODE, true, USE_E>(K, t - 1, nxt); fetch_facF<C, USE_E>(K, t - 1); }
        double h = rhs_q<C, MODE>(K, cur, musig);
        if (t < T - 1) {
            double v = pn;
            if (USE_E) {
                const double e = cur[R_H];
#pragma unroll
                for (int c = 0; c < NX; ++c) v = fma(Pn[c], rb(e, c), v);
            }
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) acc = fma(fcol[i], rb(v, i), acc);
            h += acc;
        }
#pragma unroll
        for (int j = NX; j < NT; ++j) {
            const double hj = rb(h, j) * rb(rd, j);
            const double lij = (r < NT && (r < NX || r > j)) ? lcol[j - NX] : 0.0;
            h = (r == j) ? hj : fma(-lij, hj, h);
        }
        // the multipliers are linear in the right-hand side and only their sum over the affine and the corrector
        // solve is used (the step of y): the corrector leaves p_aff + p_cor, and the forward sweeps multiply
        // P_t once, by the summed dx (no P_t in the affine sweep: a quarter of the factor reads)
        if (K.xl) w[L.PV + t * NX + r] = (MODE == CORRECTOR) ? h + cur[R_A] : h;
        if (K.ul) w[L.YB + t * NU + K.a] = h;
        pn = K.xl ? h : 0.0;
        if (USE_E) {
#pragma unroll
            for (int c = 0; c < NX; ++c) Pn[c] = K.xl ? prow[c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < R_N; ++i) cur[i] = nxt[i];
    }
}

// forward sweep: dtau, dy, ds, dz of the solve; returns the lane-local minimum step ratio of
// (s, z) against the direction that ends up in D* (affine: this solve; corrector: affine + this)
template <class C, int MODE>
__device__ __forceinline__ double sweep_fwd(const Ctx<C> &K0, double musig)
{
    const Ctx<C> K = fresh<C>(K0);
    using S = Stage<C>;
    constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    constexpr bool USE_E = (MODE == INIT || MODE == AFFINE);
    const int r = K.r, T = K.T;
    double *w = K.w;
    const Lay &L = K.L;
    double ratio = INFINITY;
    // the knot's vectors: p_t, Luu^-1 h_u, e_t, the constraint rows of the knot's control, and in the
    // corrector the affine step they are added to
    enum { V_PV, V_YB, V_E, V_SU, V_SL, V_ZU, V_ZL, V_A, V_B, V_C, V_D, V_DXO, V_N };
    auto load_vec = [&](int t, double (&v)[V_N]) {
#pragma unroll
        for (int i = 0; i < V_N; ++i) v[i] = (i == V_SU || i == V_SL) ? 1.0 : 0.0;
        if (K.xl) {
            v[V_PV] = w[L.PV + t * NX + r];
            if (USE_E) {
                double rh[R_N];
                rhs_load<C, MODE, false, true>(K, t, rh);
                v[V_E] = rh[R_H];
            }
        }
        if (MODE == CORRECTOR && r < NT) v[V_DXO] = w[L.DX + t * NT + r];
        if (K.ul) {
            const int iu = t * NU + K.a;
            v[V_YB] = w[L.YB + iu];
            if (MODE == AFFINE || MODE == CORRECTOR) { v[V_SU] = w[L.SU + iu]; v[V_SL] = w[L.SL + iu]; v[V_ZU] = w[L.ZU + iu]; v[V_ZL] = w[L.ZL + iu]; }
            if (MODE == AFFINE) { v[V_A] = w[L.RZU + iu]; v[V_B] = w[L.RZL + iu]; }
            if (MODE == CORRECTOR) { v[V_A] = w[L.DSU + iu]; v[V_B] = w[L.DSL + iu]; v[V_C] = w[L.DZU + iu]; v[V_D] = w[L.DZL + iu]; }
        }
    };
    // dx_0 = -ry_init
    double dx = 0.0;
    if (K.xl) {
        if (MODE == INIT) dx = K.P.mx0[K.qp * NX + r];
        else if (MODE == AFFINE) dx = -w[L.RY + (T - 1) * NX + r];
    }
    double cur[V_N], nxt[V_N];
    constexpr bool WITHP = (MODE != AFFINE);        // dy comes out of the corrector sweep for both solves
    load_vec(0, cur);
    fetch_facF<C, WITHP>(K, 0);
    for (int t = 0; t < T; ++t) {
        double lrow[NU], prow[NX], frow[NT];
        wait_vm();
        const double rd = stage_L<C>(K, t, lrow);
        if (WITHP) stage_P<C>(K, t, prow);
        if (t < T - 1) S::MF::template cols<NX, 0, NT>(K.imgF(t), K.g, K.xl ? r : 0, frow);
        wait_lds();
        if (t < T - 1) { load_vec(t + 1, nxt); fetch_facF<C, WITHP>(K, t + 1); }
        if (WITHP) {   // the multiplier behind x_t:  dy_{t-1} = P_t dx_t + p_t ;  t = 0:  dy_init = -(P_0 dx_0 + p_0)
            const double dxs = (MODE == CORRECTOR) ? dx + (K.xl ? cur[V_DXO] : 0.0) : dx;        // affine + corrector
            double v = cur[V_PV];
#pragma unroll
            for (int c = 0; c < NX; ++c) v = fma(K.xl ? prow[c] : 0.0, rb(dxs, c), v);
            if (t == 0) v = -v;
            if (K.xl) w[L.DY + (t >= 1 ? t - 1 : T - 1) * NX + r] = v;
        }
#pragma unroll
        for (int b = 0; b < NU; ++b) lrow[b] = r < NT ? lrow[b] : 0.0;
        double zz = K.ul ? cur[V_YB] : 0.0;
#pragma unroll
        for (int b = 0; b < NU; ++b) {
            const double wb = row_sum(K.xl ? lrow[b] * dx : 0.0);        // (Lxu' dx)[b]
            if (r == NX + b) zz += wb;
        }
        double yv = 0.0;                                                   // Luu' yv = zz
#pragma unroll
        for (int j = NT - 1; j >= NX; --j) {
            const double s = row_sum((r > j && r < NT) ? lrow[j - NX] * yv : 0.0);
            if (r == j) yv = (zz - s) * rd;
        }
        const double dtau = K.xl ? dx : (K.ul ? -yv : 0.0);
        if (r < NT) w[L.DX + t * NT + r] = (MODE == CORRECTOR) ? cur[V_DXO] + dtau : dtau;
        if (K.ul && MODE != ADJOINT) {
            const int iu = t * NU + K.a;
            if (MODE == INIT) {            // s = ds = -rz - G dx,  z = dz = -ds   (d = 1, rs = 0)
                const double su = K.uu - dtau, sl = dtau - K.ulo;
                w[L.SU + iu] = su; w[L.SL + iu] = sl; w[L.ZU + iu] = -su; w[L.ZL + iu] = -sl;
            } else {
                const double su = cur[V_SU], sl = cur[V_SL], zu = cur[V_ZU], zl = cur[V_ZL];
                double dsu, dsl, dzu, dzl;
                // reciprocals (hardware estimate + two Newton steps, as the dense kernels' ratio test) instead
                // of IEEE divisions: ten of those per knot were most of this sweep's instructions
                const double isu = frcp(su), isl = frcp(sl);
                if (MODE == AFFINE) {
                    dsu = -cur[V_A] - dtau; dsl = -cur[V_B] + dtau;
                    dzu = -zu - zu * isu * dsu;   dzl = -zl - zl * isl * dsl;
                } else {
                    const double asu = cur[V_A], asl = cur[V_B], azu = cur[V_C], azl = cur[V_D];
                    const double rsu = (-musig + asu * azu) * isu, rsl = (-musig + asl * azl) * isl;
                    const double csu = -dtau, csl = dtau;
                    dsu = asu + csu; dsl = asl + csl;
                    dzu = azu + (-rsu - zu * isu * csu); dzl = azl + (-rsl - zl * isl * csl);
                }
                w[L.DSU + iu] = dsu; w[L.DSL + iu] = dsl; w[L.DZU + iu] = dzu; w[L.DZL + iu] = dzl;
                // get_step (batch.py:206-214, with batch_LU's dv == 0 guard): a = -v/dv where dv < 0
                ratio = fmin(ratio, dsu < 0.0 ? -su * frcp(dsu) : INFINITY);
                ratio = fmin(ratio, dsl < 0.0 ? -sl * frcp(dsl) : INFINITY);
                ratio = fmin(ratio, dzu < 0.0 ? -zu * frcp(dzu) : INFINITY);
                ratio = fmin(ratio, dzl < 0.0 ? -zl * frcp(dzl) : INFINITY);
                // DQP_FLAG_STRICT_GET_STEP: batch.py:211-214 literally -- an exactly-zero component gives -inf there
                if ((K.P.flags & DQP_FLAG_STRICT_GET_STEP) && (dsu == 0.0 || dsl == 0.0 || dzu == 0.0 || dzl == 0.0))
                    ratio = -INFINITY;
            }
        }
        if (t < T - 1) {
#pragma unroll
            for (int c = 0; c < NT; ++c) frow[c] = K.xl ? frow[c] : 0.0;
            const double dxn = mv_row<NT>(frow, dtau) + (USE_E ? cur[V_E] : 0.0);
            dx = K.xl ? dxn : 0.0;
        }
#pragma unroll
        for (int i = 0; i < V_N; ++i) cur[i] = nxt[i];
    }
    return ratio;
}

// element-wise helpers over per-QP arrays, lanes strided.  A plain `for (i = r; i < n; i += 16) dst[i] = src[i]`
// waits out one memory latency per element (nothing tells the compiler that dst and src are different
// regions of the workspace); these keep EW_DEPTH independent loads per lane in flight -- the passes
// between the sweeps were 15-20 % of an iteration at four.
constexpr int EW_DEPTH = 16;
__device__ __forceinline__ void ew_copy(double *__restrict__ dst, double *__restrict__ dst2, const double *__restrict__ src,
                                        int n, int r)
{
    int i = r;
    for (; i + 16 * (EW_DEPTH - 1) < n; i += 16 * EW_DEPTH) {
        double v[EW_DEPTH];
#pragma unroll
        for (int k = 0; k < EW_DEPTH; ++k) v[k] = src[i + 16 * k];
#pragma unroll
        for (int k = 0; k < EW_DEPTH; ++k) dst[i + 16 * k] = v[k];
        if (dst2) {
#pragma unroll
            for (int k = 0; k < EW_DEPTH; ++k) dst2[i + 16 * k] = v[k];
        }
    }
    double v[EW_DEPTH];
#pragma unroll
    for (int k = 0; k < EW_DEPTH; ++k) v[k] = i + 16 * k < n ? src[i + 16 * k] : 0.0;
#pragma unroll
    for (int k = 0; k < EW_DEPTH; ++k) {
        if (i + 16 * k < n) {
            dst[i + 16 * k] = v[k];
            if (dst2) dst2[i + 16 * k] = v[k];
        }
    }
}
__device__ __forceinline__ void ew_axpy(double *__restrict__ y, const double *__restrict__ x, double alpha, int n, int r)
{
    constexpr int D = EW_DEPTH / 2;
    int i = r;
    for (; i + 16 * (D - 1) < n; i += 16 * D) {
        double xv[D], yv[D];
#pragma unroll
        for (int k = 0; k < D; ++k) { xv[k] = x[i + 16 * k]; yv[k] = y[i + 16 * k]; }
#pragma unroll
        for (int k = 0; k < D; ++k) y[i + 16 * k] = fma(alpha, xv[k], yv[k]);
    }
    double xv[D], yv[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { const bool in = i + 16 * k < n; xv[k] = in ? x[i + 16 * k] : 0.0; yv[k] = in ? y[i + 16 * k] : 0.0; }
#pragma unroll
    for (int k = 0; k < D; ++k)
        if (i + 16 * k < n) y[i + 16 * k] = fma(alpha, xv[k], yv[k]);
}

// the iterate (X, Y, SU, SL, ZU, ZL: contiguous at the head of the layout) -> the best-iterate arrays, or,
// under the batch rule, -> this iteration's snapshot only: the snapshots are then the best-iterate store
// (the outputs are read back from the best one, the finish pass picks an earlier one where the rule says so)
template <class C>
__device__ __forceinline__ void copy_best(const Ctx<C> &K, double *snap)
{
    constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    double *w = K.w;
    const Lay &L = K.L;
    const int T = K.T, r = K.r;
    const int len = T * (NT + NX + 4 * NU);
    if (snap) {
        if (K.live) ew_copy(snap, nullptr, w + L.X, len, r);
    } else {
        ew_copy(w + L.BX, nullptr, w + L.X, len, r);      // BX, BY, BSU, BSL, BZU, BZL mirror X .. ZL
    }
}

#ifndef DQP_RIC_WPE
#define DQP_RIC_WPE 2
#endif
// DYN: 0 the equality residual is A z - b; 1 a registered model's own step (model_residuals); 2 supplied by the
// caller, one iteration range [P.itBegin, P.itEnd) per launch, the scalar state of the loop parked behind the workspaces
// (dqp_mpc_qp_forward_stepped)
enum { RES_LINEAR = 0, RES_MODEL = 1, RES_CALLER = 2 };
constexpr int STEP_STATE = 8;      // doubles per problem slot
template <class C, int DYN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DQP_RIC_WPE, DQP_RIC_WPE))) void forward_kernel(KParams P, int T)
{
    constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    const int lane = threadIdx.x, r = lane & 15;
    long long qp = (long long)blockIdx.x * 4 + (lane >> 4);
    bool live = qp < P.B;
    if (!live) qp = P.B - 1;
    int maxIter = P.maxIter;
    const bool batch = (P.flags & DQP_FLAG_BATCH_TERMINATION) != 0;
    if (P.cap) {        // pass 2 of the batch rule: only the listed QPs, up to the reference's stop
        maxIter = min(maxIter, P.cap[0]);
        live = live && term_flagged(P, qp);
        if (__builtin_amdgcn_ballot_w64(live) == 0) return;
    }
    term_zero_acc(P);
    const Lay L = layout(NX, NU, T);
    // a duplicated (padding) row works on its own copy of the last problem's scratch: rows must not race
    const long long slot = (long long)blockIdx.x * 4 + (lane >> 4);
    __shared__ __attribute__((aligned(16))) double lds_img[C::WSL ? 2 : Stage<C>::TOTAL];
    using S = Stage<C>;
    double *img = C::WSL ? lds_dyn : lds_img;
    const long long qp0 = (long long)blockIdx.x * 4;
    double *ws0 = C::WSL ? lds_dyn + S::res_ws(T) : P.workspace + qp0 * (long long)L.total;
    Ctx<C> K = {P, ws0 + (lane >> 4) * (long long)L.total, L, qp, r, T, r < NX, r >= NX && r < NT, live,
                (r >= NX && r < NT) ? r - NX : 0, 0.0, 0.0,
                lane, lane >> 4, (int)min(3LL, (long long)P.B - 1 - qp0), qp0, ws0, img,
                S::res_f(T), lds_dyn + S::res_c(T), lds_dyn + S::res_fv(T)};
    K.uu = P.muu[K.a]; K.ulo = P.mul[K.a];
    if constexpr (C::WSL) {     // the whole problem into LDS, once
        S::MC::template fetch_all<NT>(img, K.Cblk(0), (long long)P.B * NT * NT * 8, NT * NT * 8, K.qmax, lane, T);
        if (T > 1) S::MF::template fetch_all<NX>(img + K.rf, K.Fblk(0), (long long)P.B * NX * NT * 8, NX * NT * 8, K.qmax, lane, T - 1);
        double *lc = lds_dyn + S::res_c(T), *lf = lds_dyn + S::res_fv(T);
        for (int i = lane; i < T * 4 * NT; i += 64) {
            const int t = i / (4 * NT), gq = (i / NT) % 4, rr = i % NT;
            lc[i] = P.mc[((long long)t * P.B + qp0 + min(gq, K.qmax)) * NT + rr];
        }
        for (int i = lane; i < (T - 1) * 4 * NX; i += 64) {
            const int t = i / (4 * NX), gq = (i / NX) % 4, rr = i % NX;
            lf[i] = P.mf[((long long)t * P.B + qp0 + min(gq, K.qmax)) * NX + rr];
        }
        wait_vm();
    }
    double *w = K.w;
    const int nineq = 2 * T * NU;
    int status = DQP_STATUS_OK;

    constexpr bool STEPPED = DYN == RES_CALLER;
    const int it0 = STEPPED ? P.itBegin : 0, it1 = STEPPED ? min(P.itEnd, maxIter) : maxIter;
    double *stp = STEPPED ? P.workspace + (long long)gridDim.x * 4 * L.total + slot * STEP_STATE : nullptr;

    // ---- initial point (batch.py:60-86)
    if (!STEPPED || it0 == 0) {
    if (!factor<C>(K, true, false)) status = DQP_STATUS_Q_NOT_PD;
    sweep_back<C, INIT>(K, 0.0);
    sweep_fwd<C, INIT>(K, 0.0);
    {
        ew_copy(w + L.X, nullptr, w + L.DX, T * NT, r);
        ew_copy(w + L.Y, nullptr, w + L.DY, T * NX, r);
        double ms = INFINITY, mz = INFINITY;
        for (int i = r; i < T * NU; i += 16) {
            ms = fmin(ms, fmin(w[L.SU + i], w[L.SL + i]));
            mz = fmin(mz, fmin(w[L.ZU + i], w[L.ZL + i]));
        }
        ms = row_min(ms); mz = row_min(mz);
        for (int i = r; i < T * NU; i += 16) {
            if (ms < 0.0) { w[L.SU + i] -= ms - 1.0; w[L.SL + i] -= ms - 1.0; }
            if (mz < 0.0) { w[L.ZU + i] -= mz - 1.0; w[L.ZL + i] -= mz - 1.0; }
        }
    }
    }

    double best = INFINITY;
    bool have_best = false, done = false;
    int nNot = 0, iters = 0, best_it = 0;
    if (STEPPED && it0 > 0) {       // the loop's scalars as the previous launch left them
        best = stp[0];
        const long long pk = (long long)stp[1];
        have_best = pk & 1; done = pk & 2; status = (int)((pk >> 2) & 3);
        nNot = (int)stp[2]; iters = (int)stp[3]; best_it = (int)stp[4];
    }
    for (int it = it0; it < it1; ++it) {
        double nx2, nz2, ny2, sz;
        if constexpr (DYN == RES_MODEL) model_residuals<C>(K);
        if constexpr (STEPPED) {    // the caller's dyn_res(x) of this iterate, closure ordering = this layout's
            const double *ery = P.extRy + qp * (long long)(T * NX);
            for (int i = r; i < (T - 1) * NX; i += 16) w[L.RY + i] = ery[i];
        }
        const bool pd = factor_fused<C, DYN != RES_LINEAR>(K, nx2, nz2, ny2, sz);      // residuals + factorisation + affine rhs
        nx2 = row_sum(nx2); nz2 = row_sum(nz2); ny2 = row_sum(ny2); sz = row_sum(sz);
        const double mu = fabs(sz / nineq);
        const double resid = sqrt(nz2) + sqrt(ny2) + sqrt(nx2) + nineq * mu;
        if (!done) {
            iters = it + 1;
            if (!have_best || resid < best) {
                nNot = 0; have_best = true; best = resid; best_it = it;
                copy_best<C>(K, P.snap ? P.snap + ((long long)it * P.B + qp) * (long long)(T * (NT + NX + 4 * NU)) : nullptr);
            }
            else nNot += 1;
            if (batch) {
                if (P.hist && r == 0 && live) hist_put(P, qp, it, resid, mu);
                done = !(fabs(resid) < INFINITY);
            } else if ((nNot >= P.notImprovedLim && best < P.stallTol) || best < P.eps || mu > 1e32 ||
                       !(fabs(resid) < INFINITY))
                done = true;
        }
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;

        if (!pd && status == DQP_STATUS_OK) status = DQP_STATUS_Q_NOT_PD;
        // affine direction and its step (batch.py:147-163)
        double ra = row_min(sweep_fwd<C, AFFINE>(K, 0.0));
        const bool zero_step = ra == -INFINITY;     // strict get_step only
        const double alpha_a = fmin(ra, 1.0);
        double t3 = 0.0;
#pragma unroll 4
        for (int i = r; i < T * NU; i += 16) {
            t3 = fma(w[L.SU + i] + alpha_a * w[L.DSU + i], w[L.ZU + i] + alpha_a * w[L.DZU + i], t3);
            t3 = fma(w[L.SL + i] + alpha_a * w[L.DSL + i], w[L.ZL + i] + alpha_a * w[L.DZL + i], t3);
        }
        t3 = row_sum(t3);
        double sig = t3 / sz;
        sig = sig * sig * sig;
        // corrector (batch.py:165-181), step (batch.py:183-204)
        sweep_back<C, CORRECTOR>(K, mu * sig);
        const double rc = row_min(sweep_fwd<C, CORRECTOR>(K, mu * sig));
        const double alpha = fmin(0.999 * rc, 1.0);
        if (zero_step || rc == -INFINITY) done = true;      // the reference's iterate is NaN from here: keep the best
        if (!done)          // X .. ZL and DX .. DZL are laid out alike: one axpy over the whole iterate
            ew_axpy(w + L.X, w + L.DX, alpha, T * (NT + NX + 4 * NU), r);
    }
    if (STEPPED) {
        if (r == 0) {
            stp[0] = best;
            stp[1] = (double)((have_best ? 1 : 0) | (done ? 2 : 0) | (status << 2));
            stp[2] = nNot; stp[3] = iters; stp[4] = best_it;
        }
        if (it1 < maxIter) {        // more iterations to come: hand the current iterate to the caller
            if (live)
                for (int i = r; i < T * NT; i += 16) P.zhat[qp * (long long)(T * NT) + i] = w[L.X + i];
            return;
        }
    }
    if (P.hist && r == 0 && live) hist_fill(P, qp, iters);
    if (!have_best) {       // max_iter == 0 cannot happen (checked on the host); kept for symmetry
        copy_best<C>(K, nullptr);
    }
    // ---- outputs in the reference's orderings
    if (live) {
        const int nz = T * NT, neq = T * NX, hm = T * NU;
        const double *bx = (P.snap && have_best) ? P.snap + ((long long)best_it * P.B + qp) * (long long)(nz + neq + 4 * hm) : w + L.BX;
        const double *by = bx + nz, *bs = by + neq, *bz = bs + 2 * hm;         // [X | Y | SU SL | ZU ZL]
        for (int i = r; i < nz; i += 16) P.zhat[qp * nz + i] = bx[i];
        for (int i = r; i < neq; i += 16) P.nu[qp * neq + i] = by[i];
        for (int i = r; i < hm; i += 16) {
            P.lam[qp * 2 * hm + i] = bz[i];       P.lam[qp * 2 * hm + hm + i] = bz[hm + i];
            P.slack[qp * 2 * hm + i] = bs[i];     P.slack[qp * 2 * hm + hm + i] = bs[hm + i];
        }
        if (r == 0) {
            if (P.info) { P.info[2 * qp] = status; P.info[2 * qp + 1] = iters; }
            if (P.best_resid) P.best_resid[qp] = best;
        }
    }
}

// Backward (qp.py:128-183; DenseQPFunction's un-clamped d with DQP_FLAG_DENSE_BACKWARD, qp.py:246-250):
// one factorisation with d = lam/slack of the returned iterate, one solve with rhs (dl/dzhat, 0, 0, 0),
// gradients scattered straight into the MPC parameters.
template <class C>
__global__ __launch_bounds__(64) void backward_kernel(KParams P, int T)
{
    constexpr int NX = C::NX, NU = C::NU, NT = C::NT;
    const int lane = threadIdx.x, r = lane & 15;
    long long qp = (long long)blockIdx.x * 4 + (lane >> 4);
    const bool live = qp < P.B;
    if (!live) qp = P.B - 1;
    const Lay L = layout(NX, NU, T);
    const long long slot = (long long)blockIdx.x * 4 + (lane >> 4);
    __shared__ __attribute__((aligned(16))) double img[Stage<C>::TOTAL];
    const long long qp0 = (long long)blockIdx.x * 4;
    Ctx<C> K = {P, P.workspace + slot * (long long)L.total, L, qp, r, T, r < NX, r >= NX && r < NT, live,
                (r >= NX && r < NT) ? r - NX : 0, 0.0, 0.0,
                lane, lane >> 4, (int)min(3LL, (long long)P.B - 1 - qp0), qp0, P.workspace + qp0 * (long long)L.total, img,
                0, nullptr, nullptr};
    double *w = K.w;
    const int nz = T * NT, neq = T * NX, hm = T * NU;
    for (int i = r; i < hm; i += 16) {
        w[L.ZU + i] = P.lamin[qp * 2 * hm + i];     w[L.ZL + i] = P.lamin[qp * 2 * hm + hm + i];
        w[L.SU + i] = P.slackin[qp * 2 * hm + i];   w[L.SL + i] = P.slackin[qp * 2 * hm + hm + i];
    }
    int status = DQP_STATUS_OK;
    if (!factor<C>(K, false, !(P.flags & DQP_FLAG_DENSE_BACKWARD))) status = DQP_STATUS_Q_NOT_PD;
    sweep_back<C, ADJOINT>(K, 0.0);
    sweep_fwd<C, ADJOINT>(K, 0.0);
    if (!live) return;
    const long long Bq = P.B;
    // dc = dx ;  dC_t = 1/2 (dx_t tau_t' + tau_t dx_t') ;  dF_t = dnu_t tau_t' + nu_t dx_t' ;  df_t = dnu_t ;
    // dx0 = -dnu_init    (qp.py:143-178 restricted to the blocks the MPC parameters occupy)
    for (int t = 0; t < T; ++t) {
        const double dxr = r < NT ? w[L.DX + t * NT + r] : 0.0;
        const double zr = r < NT ? P.zin[qp * nz + t * NT + r] : 0.0;
        if (P.mdc && r < NT) P.mdc[((long long)t * Bq + qp) * NT + r] = dxr;
        if (P.mdC) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const double v = 0.5 * (rb(dxr, i) * zr + rb(zr, i) * dxr);
                if (r < NT) P.mdC[(((long long)t * Bq + qp) * NT + i) * NT + r] = v;
            }
        }
        if (t < T - 1) {
            const double dn = K.xl ? w[L.DY + t * NX + r] : 0.0;
            const double nn = K.xl ? P.nuin[qp * neq + t * NX + r] : 0.0;
            if (P.mdF) {
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const double v = rb(dn, i) * zr + rb(nn, i) * dxr;
                    if (r < NT) P.mdF[(((long long)t * Bq + qp) * NX + i) * NT + r] = v;
                }
            }
            if (P.mdf && K.xl) P.mdf[((long long)t * Bq + qp) * NX + r] = dn;
        }
    }
    if (P.mdx0 && K.xl) P.mdx0[qp * NX + r] = -w[L.DY + (T - 1) * NX + r];
    if (r == 0 && P.info) { P.info[2 * qp] = status; P.info[2 * qp + 1] = 0; }
}

// Pass 2 of the batch-coupled rule (dqp_term.hip) when pass 1 kept its improving iterates: a flagged
// problem's outputs are rewritten from the snapshot of its best iteration before the stop -- a copy.
// One wavefront per problem; sizes at run time.
__global__ __launch_bounds__(64) void finish_kernel(KParams P, int T, int nx, int nu)
{
    const long long qp = blockIdx.x;
    if (!term_flagged(P, qp)) return;
    const int cap = min(P.maxIter, P.cap[0]);
    const double2 *h = reinterpret_cast<const double2 *>(P.histIn) + qp;
    double best = h[0].x;
    int arg = 0;
    for (int it = 1; it < cap; ++it) {
        const double v = h[(long long)it * P.B].x;
        if (v < best) { best = v; arg = it; }
    }
    const int nt = nx + nu, nz = T * nt, neq = T * nx, hm = T * nu;
    const double *sp = P.snap + ((long long)arg * P.B + qp) * (long long)(nz + neq + 4 * hm);
    const int lane = threadIdx.x;
    for (int i = lane; i < nz; i += 64) P.zhat[qp * nz + i] = sp[i];
    for (int i = lane; i < neq; i += 64) P.nu[qp * neq + i] = sp[nz + i];
    for (int i = lane; i < hm; i += 64) {
        P.slack[qp * 2 * hm + i] = sp[nz + neq + i];            P.slack[qp * 2 * hm + hm + i] = sp[nz + neq + hm + i];
        P.lam[qp * 2 * hm + i] = sp[nz + neq + 2 * hm + i];     P.lam[qp * 2 * hm + hm + i] = sp[nz + neq + 3 * hm + i];
    }
    if (lane == 0) {
        if (P.info) P.info[2 * qp + 1] = cap;
        if (P.best_resid) P.best_resid[qp] = best;
    }
}

template <class C, class Kern> int launch(Kern kernel, const KParams &P, int T, void *stream, size_t lds = 0)
{
    const int blocks = (P.B + 3) / 4;
    DQP_LAUNCH(kernel, dim3(blocks), dim3(64), lds, (hipStream_t)stream, P, T);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // namespace ric

// size table: (n_state, n_ctrl) pairs with a stage-wise kernel
#ifndef DQP_RIC_SIZES
#define DQP_RIC_SIZES                                                                                   \
    X(12, 4) X(3, 3) X(3, 1) X(4, 1) X(6, 1) X(2, 1) X(4, 2) X(5, 1) X(8, 1) X(2, 2) X(3, 2) X(6, 2) X(8, 2) X(6, 3)  \
    X(4, 4) X(8, 4) X(10, 4) X(12, 2)
#endif

bool ric_supported(int n, int m)
{
#define X(a, b) if (n == a && m == b) return true;
    DQP_RIC_SIZES
#undef X
    return false;
}

long long ric_workspace_doubles(int n, int m, int T)
{
    return ric_supported(n, m) ? ric::layout(n, m, T).total : 0;
}

int ric_forward(const KParams &P, void *stream)
{
    // the true-dynamics residual of a registered model is its own instantiation: the model's registers
    // (an RK4 step of the quadrotor) would otherwise be spilled around on the linear path too.
    // Workspace in LDS (Cfg<., ., true>) where four of them fit beside the stage without starving the CU
    // of wavefronts: up to 40 KB per wavefront, or up to 64 KB while the batch is at most two per CU.
#define X(a, b)                                                                                                       \
    if (P.mn == a && P.mm == b) {                                                                                     \
        using Cg = ric::Cfg<a, b>;                                                                                    \
        using Cl = ric::Cfg<a, b, true>;                                                                              \
        const size_t lds = (ric::Stage<Cl>::res_ws(P.mT) + 4 * (size_t)ric::layout(a, b, P.mT).total) * sizeof(double); \
        const bool wsl = !(P.flags & DQP_FLAG_RIC_GLOBAL_WS) && (lds <= 40 * 1024 || (lds <= 64 * 1024 && (P.B + 3) / 4 <= 512)); \
        if constexpr (ric::has_model<Cg>())                                                                           \
            if (P.dynId)                                                                                              \
                return wsl ? ric::launch<Cl>(ric::forward_kernel<Cl, ric::RES_MODEL>, P, P.mT, stream, lds)                     \
                           : ric::launch<Cg>(ric::forward_kernel<Cg, ric::RES_MODEL>, P, P.mT, stream);                         \
        if (P.dynId) return 1;                                                                                        \
        return wsl ? ric::launch<Cl>(ric::forward_kernel<Cl, ric::RES_LINEAR>, P, P.mT, stream, lds)                            \
                   : ric::launch<Cg>(ric::forward_kernel<Cg, ric::RES_LINEAR>, P, P.mT, stream);                                \
    }
    DQP_RIC_SIZES
#undef X
    return 1;
}

// caller-driven iterations (dqp_mpc_qp_forward_stepped): always the caller's workspace, STEP_STATE doubles per slot behind it
long long ric_stepped_workspace_doubles(int n, int m, int T, int B)
{
    if (!ric_supported(n, m)) return 0;
    return (long long)((B + 3) / 4 * 4) * (ric::layout(n, m, T).total + ric::STEP_STATE);
}
int ric_forward_stepped(const KParams &P, void *stream)
{
#define X(a, b) if (P.mn == a && P.mm == b) return ric::launch<ric::Cfg<a, b>>(ric::forward_kernel<ric::Cfg<a, b>, ric::RES_CALLER>, P, P.mT, stream);
    DQP_RIC_SIZES
#undef X
    return 1;
}

int ric_snapshot_doubles(int n, int m, int T) { return ric_supported(n, m) ? T * (2 * n + 5 * m) : 0; }

int ric_finish(const KParams &P, void *stream)
{
    DQP_LAUNCH(ric::finish_kernel, dim3(P.B), dim3(64), 0, (hipStream_t)stream, P, P.mT, P.mn, P.mm);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

int ric_backward(const KParams &P, void *stream)
{
#define X(a, b) if (P.mn == a && P.mm == b) return ric::launch<ric::Cfg<a, b>>(ric::backward_kernel<ric::Cfg<a, b>>, P, P.mT, stream);
    DQP_RIC_SIZES
#undef X
    return 1;
}

}  // namespace dqp
