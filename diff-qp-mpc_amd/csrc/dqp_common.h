// dqp_common.h -- kernel parameter block shared by the HIP translation units of libdqp_hip.so
#ifndef DQP_COMMON_H_
#define DQP_COMMON_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dqp.h"
#include "dqp_trace.h"

namespace dqp {

constexpr int WAVE = 64;

struct KParams {
    const double *Q, *p, *G, *h, *A, *b;
    long long sQ, sp, sG, sh, sA, sb;
    // forward outputs
    double *zhat, *lam, *nu, *slack, *best_resid;
    // backward inputs / outputs
    const double *zin, *lamin, *nuin, *slackin, *gin;
    double *dQ, *dp, *dG, *dh, *dA, *db;
    int32_t *info;
    double *workspace;            // caller-provided scratch (dqp_workspace_bytes), may be NULL
    unsigned long long *stamps;   // diagnostic: s_memtime at phase boundaries (16 per workgroup) or NULL
    int B, N, M, E;
    int ldz, ldm, lde, ldt;
    double eps, stallTol;
    int maxIter, notImprovedLim;
    unsigned flags;
    // batch-coupled termination (DQP_FLAG_BATCH_TERMINATION, dqp_term.hip)
    double *hist;          // pass 1: (B, histIters, 2) = (resid, mu) per iteration, or NULL
    const int32_t *cap;    // pass 2: cap[0] = iteration cap I*, cap[TERM_HDR + qp] = iteration of this QP's best iterate
    int histIters;
    // true-dynamics equality residual (dqp_opts.dyn_*): registered model evaluated per iteration
    int dynId, dynT, dynN, dynM;
    double dynDt;
    const double *dynX0;
    // MPC-structured I/O (dqp_mpc_qp_forward / _backward, null-space kernels only): the dense
    // (Q,p,G,h,A,b) and their gradients are never materialised in HBM.  mC != NULL selects it.
    const double *mC, *mc, *mF, *mf, *mx0, *mul, *muu;     // time-major inputs (qp_wrapper layout)
    double *mdC, *mdc, *mdF, *mdf, *mdx0;                  // backward outputs
    int mn, mm, mT;
    // batch rule, null-space kernels: improving iterates of pass 1 ([it][B][snapDim]) and the history
    // as the finish pass reads it
    double *snap;
    const double *histIn;
    // caller-driven iterations (dqp_mpc_qp_forward_stepped, stage-wise kernels): iterations itBegin .. itEnd-1 of
    // maxIter, the equality residual of iteration itBegin read from extRy (B, T n) in the closure's ordering
    const double *extRy;
    int itBegin, itEnd;
};

constexpr int TERM_HDR = 8;   // int32 header words in front of the per-problem best-iteration list

// Termination buffer (dqp_termination_bytes), see dqp_term.hip: hist [histIters][B] x (resid, mu),
// then TERM_ACC_BYTES of accumulators + header, then the redo list.
constexpr int TERM_MAXIT = 64;
constexpr int TERM_ACC_BYTES = 3 * 8 + TERM_HDR * 4;

// Record one iteration of a problem's residual history (one lane per problem calls this);
// iteration-major so that the batch reduction reads it coalesced.
__device__ __forceinline__ void hist_put(const KParams &P, long long qp, int it, double resid, double mu)
{
    double2 *h = reinterpret_cast<double2 *>(P.hist) + (long long)it * P.B + qp;
    *h = make_double2(resid, mu);
}
// The problem left the loop after `iters` iterations: the rest of its history is NaN (what the
// reference's iterate is from there on: a non-finite residual never recovers, batch.py:119-131).
__device__ __forceinline__ void hist_fill(const KParams &P, long long qp, int iters)
{
    const double nan = __builtin_nan("");
    for (int it = iters; it < P.histIters; ++it)
        reinterpret_cast<double2 *>(P.hist)[(long long)it * P.B + qp] = make_double2(nan, nan);
}
// Pass 1 clears the batch reduction's accumulators itself (the reduction runs after this kernel
// on the same stream): saves a memset launch per forward call.
__device__ __forceinline__ void term_zero_acc(const KParams &P)
{
    if (P.hist && blockIdx.x == 0) {
        unsigned *a = reinterpret_cast<unsigned *>(P.hist + (long long)P.B * P.histIters * 2);
        for (int i = threadIdx.x; i < TERM_ACC_BYTES / 4; i += blockDim.x) a[i] = 0u;
    }
}

// pass 2: a problem is taken back iff its best iterate of pass 1 came at an iteration the reference never ran
__device__ __forceinline__ bool term_flagged(const KParams &P, long long qp) { return P.cap[TERM_HDR + qp] >= P.cap[0]; }

// batch rule replay (dqp_term.hip); 0 on success
size_t term_bytes(int B, int maxIter, int snapDim);
int term_decide(const KParams &P, void *term, void *stream);
int term_local_masks(const KParams &P, void *term, unsigned long long *masks, void *stream);
int term_decide_global(const KParams &P, void *term, const unsigned long long *masks, void *stream);
void term_bind_pass1(KParams &P, void *term, int snapDim);
void term_bind_pass2(KParams &P, void *term);


// DPP-row kernels (dqp_r16.hip): 4 QPs per wavefront for compile-time sizes <= 32.
// Return DQP_OK / error, or 1 when no instantiation matches (caller falls back to the
// generic kernels of dqp_pdipm.hip).
#ifdef DQP_STAMPS
extern unsigned long long *g_debug_stamps;
#endif
int r16_forward(const KParams &P, void *stream);
int r16_backward(const KParams &P, void *stream);
// null-space form of the forward kernel (dqp_r16n.hip); same return convention.  It parks
// r16n_workspace_doubles(N, M, E) doubles per QP in P.workspace (0: no instantiation).
int r16n_forward(const KParams &P, void *stream);
long long r16n_workspace_doubles(int N, int M, int E);
int r16n_snapshot_doubles(int N, int M, int E);
// stage-wise (Riccati) PDIPM for MPC-structured QPs of any horizon (dqp_ric.hip): n + m <= 16
bool ric_supported(int n_state, int n_ctrl);
long long ric_workspace_doubles(int n_state, int n_ctrl, int T);
int ric_forward(const KParams &P, void *stream);      // 1: no kernel for this (n, m)
long long ric_stepped_workspace_doubles(int n_state, int n_ctrl, int T, int B);
int ric_forward_stepped(const KParams &P, void *stream);   // iterations [P.itBegin, P.itEnd), residual from P.extRy
int ric_backward(const KParams &P, void *stream);
int ric_snapshot_doubles(int n_state, int n_ctrl, int T);   // iterate snapshot of the batch rule's finish pass
int ric_finish(const KParams &P, void *stream);
// dense QPs above DQP_MAX_DIM (dqp_big.hip): one QP per workgroup, matrices in the workspace, MFMA tiles
long long big_workspace_doubles(int N, int M, int E);
bool big_fits(int N, int M, int E);         // the solver's vectors fit the LDS of a CU
int big_forward(const KParams &P, void *stream);
int big_backward(const KParams &P, void *stream);

// dqp_al_banded.hip: dqp_al_banded_newton_step with `keep` = does the caller use this step's factor afterwards
int al_banded_newton_step_keep(const dqp_al_mpc_dims *d, int dyn_id, double dt, const double *xu, const double *x0,
                               const double *Qdiag, const double *q, const double *lam, const double *rho,
                               const double *u_lower, const double *u_upper, double *update, void *factor, int32_t *info,
                               void *stream, int keep);
// backward restarted from the context r16n_forward left in P.workspace (DQP_FLAG_BACKWARD_CTX)
int r16n_backward(const KParams &P, void *stream);

}  // namespace dqp
#endif
