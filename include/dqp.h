/*
 * dqp.h -- C ABI of the MI355X-native differentiable batched QP solver (libdqp_hip.so).
 *
 * This is the drop-in boundary for the hot path of swami1995/diff-qp-mpc.  The reference
 * has no C/FFI layer on this path (its boundary is a Python torch.autograd.Function built by
 * a factory closure); each entry point below names the reference interface it replaces, and
 * INTEGRATION.md shows the ~100-line Python binding a maintainer of the reference would add.
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HIP), fp64, row-major, contiguous inside one batch
 *     element.  The caller owns every buffer; the library allocates nothing and keeps no state.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls only enqueue
 *     work; they never synchronise, allocate or copy, so they are hipGraph-capturable.
 *   - A batch stride of 0 means "this parameter is shared by all batch elements" (the
 *     reference's expandParam rule, qpth/util.py:36-43).
 *   - Return value: DQP_OK or a negative DQP_ERR_* for argument errors detected on the host.
 *     Per-problem numerical status is written to `info` on the device (see below).
 *   - Sizes: this build solves QPs with nz, nineq, neq <= DQP_MAX_DIM on the fused
 *     one-wavefront-per-QP kernels; larger problems return DQP_ERR_TOO_LARGE.
 */
#ifndef DQP_H_
#define DQP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DQP_VERSION 300 /* 0.3.0: DQP_FLAG_STRICT_GET_STEP, dqp_mpc_qp_forward_stepped (caller-supplied equality
                           residual, one PDIPM iteration range per call), dqp_trace_begin / dqp_trace_end, DQP_MAX_DIM_LARGE
                           (blocked dense kernels above DQP_MAX_DIM);
                           0.2.1x: dqp_mpc_qp_backward takes C and F, dqp_mpc_dims.dyn_id, dqp_mpc_qp_termination_bytes,
                           dqp_term_local_masks + dqp_qp_forward_finish,
                           dqp_al_newton_solve_bytes(dims, banded); 212: DQP_FLAG_RIC_GLOBAL_WS, smaller
                           dqp_mpc_qp_workspace_bytes of the stage-wise kernels */
#define DQP_MAX_DIM 64
#define DQP_MAX_DIM_LARGE 512 /* dqp_qp_forward / dqp_qp_backward above DQP_MAX_DIM: one QP per workgroup, matrices
                                 blocked through LDS (csrc/dqp_big.hip); the reference's prof-linear.py sizes
                                 (nz = nineq up to 500, prof-linear.py:38-46) */

enum {
    DQP_OK = 0,
    DQP_ERR_BAD_ARG = -1,    /* null pointer, negative size, nineq == 0 ...              */
    DQP_ERR_TOO_LARGE = -2,  /* a dimension exceeds DQP_MAX_DIM or LDS capacity           */
    DQP_ERR_LAUNCH = -3,     /* hipLaunchKernel / hipFuncSetAttribute reported an error   */
    DQP_ERR_NO_DEVICE = -4
};

/* per-problem status written to info[2*i + 0] */
enum {
    DQP_STATUS_OK = 0,
    DQP_STATUS_Q_NOT_PD = 1,    /* reference: RuntimeError('Q is not SPD.') qp.py:86 /
                                   LU(Q) failure batch.py:381-388                          */
    DQP_STATUS_A_RANK_DEF = 2   /* A Q^-1 A^T not positive definite (A rank deficient)    */
};

/* flags */
#define DQP_FLAG_DENSE_BACKWARD 1u /* backward of DenseQPFunction (qp.py:239-270): d = lam/slack
                                      without the 1e-8 clamps of QPFunction (qp.py:149)   */

#define DQP_FLAG_GENERIC_ONLY 2u   /* testing: skip the size-specialised DPP-row kernels     */
#define DQP_FLAG_BACKWARD_CTX 8u   /* dqp_qp_backward: `workspace` is the buffer dqp_qp_forward filled
                                      for the SAME Q, G, A (it holds the factorisations, like the
                                      reference's ctx.Q_LU / S_LU / R, qp.py:93-95): skip the
                                      refactorisation.  Ignored where no such kernel exists.   */
#define DQP_FLAG_BATCH_TERMINATION 16u /* forward: the reference's batch-coupled stopping rule
                                      (batch.py:119-144), exactly: every problem is iterated to
                                      max_iter while its (resid, mu) history is recorded in
                                      `termination`, a reduction over the batch replays the
                                      reference's rule (no sample improved for not_improved_lim
                                      iterations | best_resids.max() < eps | mu.min() > 1e32) to
                                      find the iteration the reference stops at, and the problems
                                      whose best iterate came later are taken back to there: the
                                      null-space kernels keep every improving iterate of pass 1 in
                                      `termination` and finish from the right one (an epilogue, no
                                      second solve); the other kernels re-solve up to there.  Needs
                                      the `termination` buffer (dqp_termination_bytes).  Without the
                                      flag a problem stops on its own (see dqp_qp_forward).        */
#define DQP_FLAG_HISTORY_ONLY 32u   /* with DQP_FLAG_BATCH_TERMINATION: stop after pass 1 (every problem
                                      iterated to max_iter, (resid, mu) history recorded in
                                      `termination` as (max_iter, nbatch, 2) doubles, best iterate over
                                      all max_iter iterations returned): diagnostics and profiling of
                                      the dominant launch on its own                                */
#define DQP_FLAG_NO_NULLSPACE 4u   /* forward: keep the equality rows in the iteration even when a
                                      workspace is given (the kernel used without one)       */
#define DQP_FLAG_RIC_GLOBAL_WS 64u  /* testing: dqp_mpc_qp_forward's stage-wise kernels keep their iterates
                                      and factors in the caller's workspace even where they would
                                      fit in LDS (the path long horizons take)                  */

#define DQP_FLAG_STRICT_GET_STEP 128u /* forward: batch.py:211-214 literally.  The reference's QPFunction solver computes the
                                      step ratios as -v/dv with no guard: a step component that is exactly 0.0 gives
                                      -inf, alpha = -inf, the iterate turns NaN and the problem keeps the best iterate
                                      it had (its history is NaN from there on).  Its DenseQPFunction solver guards
                                      the division (batch_LU.py:203-214), and so do these kernels by default.  With
                                      this flag a problem whose affine or combined step has an exactly-zero slack or
                                      multiplier component is frozen the way the reference's NaN freezes it.  Which
                                      problems meet an exact zero depends on the arithmetic order, so this reproduces
                                      the reference's behaviour class, not its bit pattern.       */

#define DQP_FLAG_STAGEWISE 256u /* dqp_mpc_qp_forward / _backward: take the stage-wise (Riccati) kernels even where a
                                      null-space instantiation exists for the shape (testing; and the backward of a
                                      dqp_mpc_qp_forward_stepped solve, whose workspace is the stage-wise one)        */

typedef struct dqp_dims {
    int32_t nbatch;
    int32_t nz;
    int32_t nineq;
    int32_t neq;       /* 0 = no equality constraints (A, b may be NULL)                   */
    /* batch strides in ELEMENTS (doubles); 0 = shared across the batch                    */
    int64_t stride_Q, stride_p, stride_G, stride_h, stride_A, stride_b;
} dqp_dims;

typedef struct dqp_opts {
    double eps;               /* qp.py:19  eps=1e-12                                       */
    double stall_tol;         /* per-problem early exit needs best_resid < stall_tol (1e-10) */
    int32_t max_iter;         /* qp.py:20  maxIter=20                                      */
    int32_t not_improved_lim; /* qp.py:19  notImprovedLim=3                                */
    uint32_t flags;
    int32_t reserved;
    /* True-dynamics equality residual (qp_wrapper.py:309,316,326-345 -> batch_LU.py:97 / batch.py:102):
     * with dyn_id != 0 the PDIPM evaluates ry = dyn_res(z) = [f(x_t,u_t) - x_{t+1}]_{t<T-1}, x_0 - x0
     * with f the registered device model (DQP_DYN_*, below) instead of A z - b, every iteration, as
     * the reference does with its Python closure; z is per knot [x_t (n_state), u_t (n_ctrl)], so
     * nz = dyn_T (n_state + n_ctrl) and neq = dyn_T n_state (+ n_state rows x_{T-1} = 0 of
     * add_goal_constraint, qp_wrapper.py:339-341).  A and b (the linearisation) still
     * define the Newton systems and the starting point.  dyn_x0: device pointer (nbatch, n_state).
     * Runs on the generic one-QP-per-wavefront kernels.  dyn_id == 0: ry = A z - b.                 */
    int32_t dyn_id;
    int32_t dyn_T;
    double dyn_dt;
    const double *dyn_x0;
} dqp_opts;

int dqp_version(void);
const char *dqp_error_string(int code);

/* Bytes of caller-provided device workspace for dqp_qp_forward (8-byte aligned).  Optional:
 * with workspace == NULL, or a size for which this returns 0, every solver state lives in
 * LDS/registers.  With it, the size-specialised forward kernels eliminate the equality
 * constraints once (null-space form), park the elimination's reflectors there between setup and
 * the final back-transformation -- ~1.4x faster at the metric size, same iterates in exact
 * arithmetic -- and leave the factorisation context (Lq, reflectors, [Gz | W], U, and the particular
 * solution of the equality rows; ~16 KB per QP at the metric size) for dqp_qp_backward and for the
 * finish pass of the batch rule: pass the same buffer with DQP_FLAG_BACKWARD_CTX to skip
 * the refactorisation (2x faster backward).  Without that flag backward needs no workspace. */
size_t dqp_workspace_bytes(const dqp_dims *dims);

/* Bytes of the device buffer DQP_FLAG_BATCH_TERMINATION needs (per-iteration residual history,
 * the batch reduction's accumulators, each problem's best-iteration index and -- sizes served by the null-space kernels --
 * max_iter iterate snapshots of (nz - neq) + 2 nineq + 2 doubles per problem: 12 KB per QP at the
 * metric size); 0 without the flag.  max_iter <= 64. */
size_t dqp_termination_bytes(const dqp_dims *dims, const dqp_opts *opts);

/*
 * The batch-coupled rule over a batch that is SHARDED across devices (the reference's rule couples every
 * sample of the batch it is given, batch.py:119-144; shards of one logical batch can reproduce the stop of
 * the whole batch with one 24-byte exchange):
 *   1. dqp_qp_forward(flags | DQP_FLAG_BATCH_TERMINATION | DQP_FLAG_HISTORY_ONLY) on every shard;
 *   2. dqp_term_local_masks -> masks[3] (device memory): bit `it` of masks[0] = some problem of the shard
 *      improved at iteration it, of masks[1] = some problem has not best_resid < eps, of masks[2] = some problem
 *      has not mu > 1e32;
 *   3. the caller ORs the masks over the shards (e.g. all_reduce(op=BOR) on an int64 tensor);
 *   4. dqp_qp_forward_finish with the combined masks: the rule, then pass 2 of this shard.
 * With the masks of a single shard step 2-4 equal what dqp_qp_forward does in one call.
 */
int dqp_term_local_masks(const dqp_dims *dims, const dqp_opts *opts, void *termination, uint64_t *masks,
                         void *stream);
int dqp_qp_forward_finish(const dqp_dims *dims, const dqp_opts *opts, const double *Q, const double *p,
                          const double *G, const double *h, const double *A, const double *b, double *zhat,
                          double *lam, double *nu, double *slack, int32_t *info, double *best_resid,
                          void *workspace, void *termination, const uint64_t *masks, void *stream);

/*
 * Replaces: qpth.qp.QPFunction(...).forward  (qpth/qp.py:24-126) =
 *           pdipm_b.pre_factor_kkt + pdipm_b.forward (qpth/solvers/pdipm/batch.py:377-428,
 *           46-208), with dyn_res(x) = A x - b and cost_grad(x) = Q x + p; and the forward of
 *           qpth.qp.DenseQPFunction (qpth/qp.py:219-237 + batch_LU.py:29-201), which solves the
 *           same Newton systems through a regularised full-KKT LU.
 * Outputs: zhat (B,nz); lam (B,nineq), nu (B,neq), slack (B,nineq) -- what the reference
 *          stashes on ctx for backward (qp.py:95,125); info (B,2) int32 = {status, PDIPM
 *          iterations run}; best_resid (B) or NULL.
 * Termination: with DQP_FLAG_BATCH_TERMINATION the reference's batch-coupled rule is reproduced
 * exactly (what the Python mirrors use by default).  Without it, termination is per problem
 * (faster in batches larger than the GPU holds at once, float-tolerance parity): the reference's rule is batch-coupled (batch.py:127-144: stop
 * when NO sample improved for not_improved_lim consecutive iterations, so in a large batch
 * every sample effectively runs max_iter iterations).  Here a problem stops when (a) it has
 * not improved for not_improved_lim consecutive iterations AND its best residual is already
 * below stall_tol (converged, stagnating at round-off), (b) its best residual < eps / 10 (one
 * Newton step past the reference's threshold: in the reference every problem of a large batch
 * keeps iterating past eps, and the extra decade is what keeps the duals of weakly active
 * constraints -- and the gradients that depend on lam/slack -- within tolerance of it), (c) its
 * residual is no longer finite (the iterate can never recover; the reference keeps spinning
 * on NaNs), (d) mu > 1e32, or (e) after max_iter iterations.  The best-residual iterate is
 * returned, as in batch.py:119-140,208.
 */
int dqp_qp_forward(const dqp_dims *dims, const dqp_opts *opts,
                   const double *Q, const double *p, const double *G, const double *h,
                   const double *A, const double *b,
                   double *zhat, double *lam, double *nu, double *slack,
                   int32_t *info, double *best_resid,
                   void *workspace, void *termination, void *stream);

/*
 * Replaces: QPFunctionFn.backward (qpth/qp.py:128-183) = factor_kkt + solve_kkt
 *           (batch.py:434-469, 351-374) with rhs (dl_dzhat, 0, 0, 0) and the outer-product
 *           gradient formulas; with DQP_FLAG_DENSE_BACKWARD, DenseQPFunction's Solver.backward
 *           (qp.py:239-270).
 * Writes PER-SAMPLE gradients dQ (B,nz,nz) dp (B,nz) dG (B,nineq,nz) dh (B,nineq)
 * dA (B,neq,nz) db (B,neq); the caller applies .mean(0) for parameters that were shared
 * (qp.py:160-178).  Any gradient pointer may be NULL to skip it.
 */
int dqp_qp_backward(const dqp_dims *dims, const dqp_opts *opts,
                    const double *Q, const double *G, const double *A,
                    const double *zhat, const double *lam, const double *nu,
                    const double *slack, const double *dl_dzhat,
                    double *dQ, double *dp, double *dG, double *dh, double *dA, double *db,
                    int32_t *info, void *workspace, void *stream);

/* ------------------------------------------------------------------ MPC-structured QPs */

typedef struct dqp_mpc_dims {
    int32_t nbatch;
    int32_t n_state;
    int32_t n_ctrl;
    int32_t T;            /* horizon; nz = T (n_state+n_ctrl), neq = T n_state                */
    int32_t has_bounds;   /* 1: nineq = 2 T n_ctrl (box on u); 0: nineq = n_ctrl placeholder  */
    int32_t dyn_id;       /* dqp_mpc_qp_*: 0, or a registered model (DQP_DYN_*) whose true step replaces
                             F tau + f in the equality residual of the PDIPM iterations -- the
                             reference's dyn_res closure (qp_wrapper.py:309,316); step = dqp_opts.dyn_dt.
                             Served by the stage-wise kernels.  Ignored by dqp_mpc_assemble etc.       */
} dqp_mpc_dims;

/*
 * Replaces: qp_wrapper.MPC.compute_Qq_dense / compute_Ab_dense / compute_Gh_dense
 *           (qpth/qp_wrapper.py:638-679) as called from single_qp (qp_wrapper.py:311-313).
 * Inputs are the reference's time-major tensors C (T,B,nt,nt) c (T,B,nt) F (T-1,B,n,nt)
 * f (T-1,B,n) x0 (B,n) and the control bounds u_lower/u_upper (n_ctrl,) (ignored without
 * bounds).  Outputs are the dense batch-major QP (Q,p,G,h,A,b) that dqp_qp_forward consumes.
 */
int dqp_mpc_assemble(const dqp_mpc_dims *dims, const double *C, const double *c, const double *F,
                     const double *f, const double *x0, const double *u_lower,
                     const double *u_upper, double *Q, double *p, double *G, double *h,
                     double *A, double *b, void *stream);

/*
 * Adjoint of dqp_mpc_assemble (what autograd derives from the index scatters of
 * qp_wrapper.py:644-679): gathers dC,dc,dF,df,dx0 from dQ,dp,dA,db.  Any pointer may be NULL
 * (missing inputs read as zero, missing outputs are skipped).
 */
int dqp_mpc_assemble_backward(const dqp_mpc_dims *dims, const double *dQ, const double *dp,
                              const double *dA, const double *db, double *dC, double *dc,
                              double *dF, double *df, double *dx0, void *stream);

/*
 * The MPC QP end to end without the dense detour: what qp_wrapper.MPC.single_qp does between
 * linearize_dynamics and the line search (qp_wrapper.py:311-319: compute_Qq/Ab/Gh_dense +
 * qp.DenseQPFunction()) in ONE solve launch -- the kernel builds the rows of (Q,p,G,h,A,b) in
 * registers from (C,c,F,f,x0,bounds) and solves; backward scatters straight into
 * (dC,dc,dF,df,dx0) (the adjoint of the assembly applied on chip).  HBM traffic per QP at
 * n=3 m=3 T=5: 0.7 k doubles in instead of 2.3 k, 0.3 k gradient doubles out instead of 2.3 k.
 * Needs control bounds, a supported shape (dqp_mpc_qp_supported) and the workspace
 * (dqp_mpc_qp_workspace_bytes).  Two kernel families serve it:
 *   - QP sizes with a null-space kernel (small horizons, nz <= 48): the whole QP lives in registers, the
 *     workspace carries the factorisation context from forward to backward (C, F unused by backward);
 *   - any other horizon with n_state + n_ctrl <= 16 and a compiled (n_state, n_ctrl) pair -- e.g.
 *     BASELINE config 4: n 12, m 4, T 30, nz 480 --: stage-wise PDIPM, every KKT solve a Riccati
 *     recursion over the knots (O(T (n+m)^3)); iterates and per-knot factors stream through the
 *     workspace; backward refactors at the returned iterate and needs C and F again.
 * tau (B, T, n_state+n_ctrl) = the QP solution per knot [x_t, u_t]; lam/nu/slack/info/best_resid and
 * the termination modes as in dqp_qp_forward; backward = DenseQPFunction's (un-clamped d).
 */
int dqp_mpc_qp_supported(const dqp_mpc_dims *dims);
size_t dqp_mpc_qp_workspace_bytes(const dqp_mpc_dims *dims);
/* the `termination` buffer of dqp_mpc_qp_forward under DQP_FLAG_BATCH_TERMINATION (history, best-iteration list and the
 * improving iterates of pass 1, from which pass 2 finishes without solving again); 0 without the flag */
size_t dqp_mpc_qp_termination_bytes(const dqp_mpc_dims *dims, const dqp_opts *opts);
int dqp_mpc_qp_forward(const dqp_mpc_dims *dims, const dqp_opts *opts, const double *C, const double *c,
                       const double *F, const double *f, const double *x0, const double *u_lower,
                       const double *u_upper, double *tau, double *lam, double *nu, double *slack,
                       int32_t *info, double *best_resid, void *workspace, void *termination,
                       void *stream);
int dqp_mpc_qp_backward(const dqp_mpc_dims *dims, const dqp_opts *opts, const double *C, const double *F,
                        const double *tau, const double *lam, const double *nu, const double *slack,
                        const double *dl_dtau, double *dC, double *dc, double *dF, double *df,
                        double *dx0, int32_t *info, void *workspace, void *stream);

/*
 * dqp_mpc_qp_forward for a dynamics model the library cannot evaluate (a caller's torch module):
 * the reference passes its PDIPM a Python closure for the equality residual and calls it once per iteration on
 * the current iterate (qp_wrapper.py:309,316 -> batch_LU.py:97: ry = dyn_res(x)).  Here the caller drives the
 * iterations: a call runs ONE PDIPM iteration (it_end == it_begin + 1) of the stage-wise kernels with
 * `ext_ry` (B, T n_state) as the equality residual of the current iterate, in the closure's ordering
 * [f(x_t,u_t) - x_{t+1} (t < T-1) ; x_0 - x0] (qp_wrapper.py:326-345; the last n_state entries are recomputed on
 * chip).  The call it_begin == it_end == 0 only computes the starting point (batch_LU.py:60-86); it_begin == 0,
 * it_end == 1 computes it and runs iteration 0 on ext_ry -- for callers that know the starting iterate.  After every call `tau`
 * holds the CURRENT iterate (B, T, n_state+n_ctrl), on which the caller evaluates its model for the next call; the
 * call with it_end == opts->max_iter finishes as dqp_mpc_qp_forward does (batch rule included) and leaves the
 * returned best iterate and its multipliers in tau, lam, nu, slack.  All solver state between calls lives in
 * `workspace` (dqp_mpc_qp_stepped_workspace_bytes: always the caller's buffer) and `termination`.
 * Shapes: compiled (n_state, n_ctrl) pairs with n_state + n_ctrl <= 16.
 */
size_t dqp_mpc_qp_stepped_workspace_bytes(const dqp_mpc_dims *dims);
size_t dqp_mpc_qp_stepped_termination_bytes(const dqp_mpc_dims *dims, const dqp_opts *opts);
int dqp_mpc_qp_forward_stepped(const dqp_mpc_dims *dims, const dqp_opts *opts, const double *C, const double *c,
                               const double *F, const double *f, const double *x0, const double *u_lower,
                               const double *u_upper, const double *ext_ry, int32_t it_begin, int32_t it_end,
                               double *tau, double *lam, double *nu, double *slack, int32_t *info,
                               double *best_resid, void *workspace, void *termination, void *stream);

/*
 * Replaces: qp_wrapper.MPC.line_search with its rollouts and cost evaluations
 * (qpth/qp_wrapper.py:417-436, 598-611, 690-692; ~60 torch ops and one host sync per round): backtracking
 * on the TRUE rollout cost, alpha in {1, decay, decay^2, ...}, per trajectory, in one launch.
 * Dynamics: dyn_id == 0 -> LinDx (F (T-1,B,n,nt), f (T-1,B,n)), else a registered device model
 * (DQP_DYN_*, step dt).  x (T,B,n), u (T,B,m): current trajectory; delta_u (T,B,m): the QP step; C, c
 * the quadratic cost.  Outputs: the last trial x_new, u_new, its cost, and alpha (B) with the
 * reference's convention (a trajectory that never improved carries one extra decay).
 * With C == NULL (then x, delta_u, c may be NULL) the call is the plain rollout of u from x0
 * (qp_wrapper.py:598-611): x_new = states, u_new = u.
 */
int dqp_mpc_line_search(const dqp_mpc_dims *dims, int dyn_id, double dt, const double *F, const double *f,
                        const double *x0, const double *x, const double *u, const double *delta_u,
                        const double *C, const double *c, double decay, int32_t max_iter,
                        double *x_new, double *u_new, double *alpha, double *cost_new, void *stream);

/*
 * Adjoint of the rollout x_{t+1} = f(x_t, u_t) (what autograd derives from qp_wrapper.py:598-611):
 * x (T,B,n) the rolled-out states, u (T,B,m), g_x (T,B,n) the cotangent of the states; outputs
 * d_x0 (B,n), d_u (T,B,m), and for LinDx (dyn_id == 0) d_F (T-1,B,n,nt), d_f (T-1,B,n).  Any output
 * may be NULL.  Registered models differentiate through the same forward-mode templates as
 * dqp_dyn_jacobian.
 */
int dqp_mpc_rollout_backward(const dqp_mpc_dims *dims, int dyn_id, double dt, const double *F,
                             const double *x, const double *u, const double *g_x, double *d_x0,
                             double *d_u, double *d_F, double *d_f, void *stream);

/* ------------------------------------------------------------ augmented-Lagrangian Newton */

typedef struct dqp_al_dims {
    int32_t nbatch;
    int32_t nz;        /* T (n_state + n_ctrl) <= 128                                      */
    int32_t ncon;      /* rows of the (clamped) constraint Jacobian: neq + nineq           */
    int32_t reserved;
} dqp_al_dims;

/*
 * Replaces, inside NewtonAL.forward (qpth/al_utils.py:403-427):
 *   merit_hess = diag(Q) + rho * Jc^T Jc                  (al_utils.py:96-102, 176-178)
 *   U, info = torch.linalg.cholesky_ex(merit_hess);  update = -cholesky_solve(grad, U)
 * Jc (B,ncon,nz) is the clamped constraint Jacobian, Qdiag (B,nz), rho (B), grad (B,nz).
 * Outputs: update (B,nz) (NaN when the factorisation fails, as the reference's NaN test at
 * al_utils.py:419 expects), L (B,nz,nz) lower Cholesky factor with a zero upper part (or NULL),
 * info (B) = 0 or the order of the first non-positive leading minor (cholesky_ex's info).
 */
int dqp_al_newton_step(const dqp_al_dims *dims, const double *Jc, const double *Qdiag,
                       const double *rho, const double *grad, double *update, double *L,
                       int32_t *info, void *stream);

/* Replaces NewtonAL.backward's  -torch.cholesky_solve(x_grad, U)  (al_utils.py:477-480):
 * out = -(L L^T)^-1 rhs. */
int dqp_al_chol_solve(const dqp_al_dims *dims, const double *L, const double *rhs, double *out,
                      void *stream);

/*
 * Replaces the Jacobian fill of al_utils.constraint_res_jac2 / dyn_res_eq_jac / dyn_res_ineq_jac
 * (qpth/al_utils.py:162-318: vmap(block_diag), zero-fills, index scatters, the active-set mask)
 * and the two bmm's J^T lam + rho Jc^T res_c of merit_grad_hessian (al_utils.py:62-102).
 * Inputs: the per-knot dynamics Jacobians Jx (B,T-1,n,n) = df/dx_t and Ju (B,T-1,n,m) = df/du_t,
 * lam (B,ncon), res_c (B,ncon) = the residual with inactive inequalities clamped to 0, rho (B).
 * Row order: (T-1) n dynamics rows x_{t+1} - f(x_t,u_t) knot-major, n rows x_0 - x0, then per
 * knot [u - u_upper (m), u_lower - u (m)]; columns per knot [x_t (n), u_t (m)];
 * ncon = T n + 2 T m, nz = T (n + m).
 * Outputs: Jc (B,ncon,nz) with the rows of inactive inequalities (res_c <= 0) zeroed (every
 * element written exactly once), gterm (B,nz) = J^T lam + rho Jc^T res_c.  Either may be NULL.
 */
typedef struct dqp_al_mpc_dims {
    int32_t nbatch;
    int32_t n_state;
    int32_t n_ctrl;
    int32_t T;
} dqp_al_mpc_dims;

int dqp_al_assemble(const dqp_al_mpc_dims *dims, const double *Jx, const double *Ju,
                    const double *lam, const double *res_c, const double *rho, double *Jc,
                    double *gterm, void *stream);

/*
 * Replaces al_utils.merit_function (qpth/al_utils.py:37-59) after the dynamics have been
 * evaluated:  merit = sum (1/2 Q xu^2 + q xu) + rho/2 |res_c|^2 + lam . res  with res the
 * residual vector in dqp_al_assemble's row order (x_{t+1} - x_next_t, x_0 - x0, u - u_upper,
 * u_lower - u) and res_c its clamped form.  ncand candidate trajectories per problem (the
 * 20-way line search of line_search_newton, al_utils.py:503-527, evaluates them as one batch)
 * share the problem's x0, Qdiag, q, lam, rho: xu (ncand,B,T,n+m), x_next (ncand,B,T-1,n) =
 * f(x_t,u_t) for the first T-1 knots, x0 (B,n), Qdiag/q (B,T,n+m), lam (B,ncon), rho (B),
 * u_lower/u_upper (m).  Output merit (ncand,B).
 */
int dqp_al_merit(const dqp_al_mpc_dims *dims, int32_t ncand, const double *xu, const double *x_next,
                 const double *x0, const double *Qdiag, const double *q, const double *lam,
                 const double *rho, const double *u_lower, const double *u_upper, double *merit,
                 void *stream);

/*
 * NewtonAL.forward for a registered device model (qpth/al_utils.py:363-456 with
 * merit_grad_hessian :62-102, constraint_res_jac2 :162-318 and line_search_newton :503-527):
 * n_steps Newton steps on the augmented Lagrangian, each = linearise the dynamics along xu
 * (forward-mode Jacobians) -> clamped constraint Jacobian + merit gradient -> Hessian (fp64 MFMA) +
 * Cholesky + solve -> merit of the 20 candidate steps 2^-k (dynamics evaluated in the kernel) ->
 * argmin / acceptance / update.  5 launches per step, nothing returns to the host in between.
 * xu (B,T,n+m) in/out; x0 (B,n); Qdiag, q (B,T,n+m); lam (B,ncon), rho (B); u_lower/u_upper (m).
 * Outputs: L (B,nz,nz) the Cholesky factor of the LAST step's Hessian (what NewtonAL.backward
 * needs, al_utils.py:458,477-480), status (B) 1.0 where the last line search accepted its step,
 * fail (one int32) != 0 when a Cholesky factorisation broke down (the reference then switches the
 * whole batch to an LU solve, al_utils.py:419-427: the caller re-runs its general path).
 * Dense form (banded == 0): n_state <= 8, n_ctrl <= 2, nz <= 128, L is (B,nz,nz).
 * Banded form (banded != 0): n_state <= 12, n_state + n_ctrl <= 16, any T (BASELINE config 4:
 * n 12, m 4, T 30, nz 480); L is the buffer of dqp_al_banded_factor_bytes (see below).
 * The workspace size depends on the form.
 */
size_t dqp_al_newton_solve_bytes(const dqp_al_mpc_dims *dims, int32_t banded);
int dqp_al_newton_solve(const dqp_al_mpc_dims *dims, int dyn_id, double dt, int32_t n_steps, int32_t banded,
                        const double *x0, const double *Qdiag, const double *q, const double *lam,
                        const double *rho, const double *u_lower, const double *u_upper, double *xu, double *L,
                        double *status, int32_t *fail, void *workspace, void *stream);

/*
 * Between two augmented-Lagrangian iterations (qpth/AL_mpc.py:296-307) for a registered device model:
 * lam_new = lam + rho res with the inequality block clamped at 0, cost (B) of the iterate
 * (diagonal cost), res_norm (B) = |clamped residual|; res in the row order of dqp_al_assemble.
 */
int dqp_al_outer_update(const dqp_al_mpc_dims *dims, int dyn_id, double dt, const double *xu, const double *x0,
                        const double *lam, const double *rho, const double *Qdiag, const double *q,
                        const double *u_lower, const double *u_upper, double *lam_new, double *cost,
                        double *res_norm, void *stream);

/*
 * Replaces: AL_mpc.MPC.al_solve for a registered device model (qpth/AL_mpc.py:254-321) as ONE call with no host
 * involvement: xu = [x_init | u_init]; cost_start; the warm start of (lam, rho) from the previous call's history
 * (al_utils.warm_start_al, al_utils.py:16-34; n_prev = 0 after reinitialize()); then al_iter times
 * [dqp_al_newton_solve (block-tridiagonal, newton_steps Newton steps with the 20-candidate line search) ->
 * lam <- clamp(lam + rho res), cost, |res_clamp|, rho <- 10 rho (AL_mpc.py:296-307)].
 * History layout, oldest first as the reference keeps it: hist_cost (al_iter+1, B), hist_lam (al_iter+1, B, ncon),
 * hist_rho (al_iter+1, B), entry 0 = the (warm-started) start, entry i+1 = after AL iteration i; the final
 * multipliers / penalty are the last entry (what the reference stores in lamda_prev / rho_prev).  prev_* = the
 * hist_* arrays of the previous call (n_prev entries).  xu (B, T, n+m) out; res_norm (B) = |res_clamp| of the last
 * iterate; factor (dqp_al_banded_factor_bytes) = the block-tridiagonal factor of the LAST Newton step, for
 * dqp_al_banded_solve (NewtonAL.backward); fail (al_iter int32): set where a Cholesky pivot was not positive in
 * that AL iteration (the reference then solves by LU: redo the call on the general path); workspace:
 * dqp_al_mpc_solve_bytes.  ncon = T n_state + 2 T n_ctrl.  n_state + n_ctrl <= 16, n_state <= 12.
 */
size_t dqp_al_mpc_solve_bytes(const dqp_al_mpc_dims *dims);
int dqp_al_mpc_solve(const dqp_al_mpc_dims *dims, int dyn_id, double dt, int32_t al_iter, int32_t newton_steps,
                     const double *x_init, const double *u_init, const double *x0, const double *Qdiag, const double *q,
                     const double *u_lower, const double *u_upper, const double *lam_in, const double *rho_in,
                     const double *prev_cost, const double *prev_lam, const double *prev_rho, int32_t n_prev,
                     double *xu, double *hist_cost, double *hist_lam, double *hist_rho, double *res_norm, double *factor,
                     double *status, int32_t *fail, void *workspace, void *stream);

/*
 * The same Newton step with the MPC structure exploited (`banded` != 0 above uses it): the Hessian
 * diag(Q) + rho Jc^T Jc is block tridiagonal in the knots, so linearisation, gradient, block Cholesky
 * and the solve run as ONE launch with every knot in registers (4 problems per wavefront), O(T (n+m)^3)
 * work, no Jacobian or Hessian in HBM.  `factor` (dqp_al_banded_factor_bytes) receives the banded
 * Cholesky factor -- per knot L_tt, 1/diag(L_tt) and L_{t+1,t}^T, in a layout private to the library (element-major
 * over the knot's lanes) -- which dqp_al_banded_solve applies for NewtonAL.backward (out = -(L L^T)^-1 rhs,
 * al_utils.py:477-480).  dqp_al_newton_solve / dqp_al_mpc_solve leave the factor of their LAST Newton step in it.
 * info (B): 0, or 1 + the knot at which a pivot was not positive.
 */
size_t dqp_al_banded_factor_bytes(const dqp_al_mpc_dims *dims, int dyn_id);
int dqp_al_banded_newton_step(const dqp_al_mpc_dims *dims, int dyn_id, double dt, const double *xu,
                              const double *x0, const double *Qdiag, const double *q, const double *lam,
                              const double *rho, const double *u_lower, const double *u_upper,
                              double *update, void *factor, int32_t *info, void *stream);
int dqp_al_banded_solve(const dqp_al_mpc_dims *dims, int dyn_id, const void *factor, const double *rhs,
                        double *out, void *stream);
/*
 * The same step for a dynamics the CALLER linearised -- a torch module with its own Jacobians such as
 * deqmpc/envs.py:50-82 or RexQuadrotor_dynamics_jac (rex_quadrotor.py:131-146) -- at any horizon:
 * x_next (B,T-1,n) = f(x_t,u_t), Jx (B,T-1,n,n) = df/dx, Ju (B,T-1,n,m) = df/du instead of a registered
 * model.  Lifts the nz <= 128 limit of dqp_al_newton_step for user dynamics (config 4 with the reference's
 * own quadrotor module: nz = 480).  Compiled (n_state, n_ctrl) pairs: DQP_BAND_SIZES in csrc/dqp_al_banded.hip;
 * others return DQP_ERR_TOO_LARGE.  The factor has the layout of dqp_al_banded_factor_bytes(dims, 0) and is
 * applied by dqp_al_banded_solve(dims, 0, ...).
 */
int dqp_al_banded_newton_step_jac(const dqp_al_mpc_dims *dims, const double *xu, const double *x0,
                                  const double *Qdiag, const double *q, const double *lam, const double *rho,
                                  const double *u_lower, const double *u_upper, const double *x_next,
                                  const double *Jx, const double *Ju, double *update, void *factor,
                                  int32_t *info, void *stream);

/* Testing / A-B runs: pins how many lanes a problem of the block-tridiagonal kernels occupies (8: two problems per
 * 16-lane DPP row where n_state + n_ctrl <= 8; 16: one per row; 0: chosen from the batch size and the device's CU
 * count, the default).  The environment variable DQP_AL_LANE_GROUP=8|16 sets the initial value (read once). */
int dqp_al_lane_group(int width);

/* ----------------------------------------------------------------- device dynamics registry */

/*
 * Robots and pendulum models the reference evaluates through per-robot torch extensions or Python
 * modules, available to callers (and inlined by the MPC / AL kernels) as device code:
 *   DQP_DYN_PENDULUM1L / CARTPOLE1L / CARTPOLE2L  deqmpc/my_envs/{pendulum1l,cartpole1l,cartpole2l}
 *       (CasADi RK4 step of the rigid-body model; state x = [q, qdot], n_state = 2 nq = 2 / 4 / 6,
 *       the control drives joint 0: deqmpc/my_envs/dynamics.py:26-63)
 *   DQP_DYN_PENDULUM_EULER   deqmpc/envs.py:5-47 PendulumDynamics (n_state 2, semi-implicit Euler)
 *   DQP_DYN_PENDULUM_DX      qpth/env_dx/pendulum.py:18-83 PendulumDx, simple=True, default
 *       parameters (n_state 3: cos th, sin th, thdot; control clamped to +-2)
 *   DQP_DYN_REXQUADROTOR     deqmpc/rex_quadrotor.py:7-129 RexQuadrotor_dynamics, default parameters
 *       (BASELINE config 4: n_state 12 = position, MRP attitude, body velocity, body rate;
 *       n_ctrl 4 motor commands; RK4)
 */
enum {
    DQP_DYN_PENDULUM1L = 1,
    DQP_DYN_CARTPOLE1L = 2,
    DQP_DYN_CARTPOLE2L = 3,
    DQP_DYN_PENDULUM_EULER = 4,
    DQP_DYN_PENDULUM_DX = 5,
    DQP_DYN_REXQUADROTOR = 6
};

/* n_state / n_ctrl of a registered model; DQP_ERR_BAD_ARG for an unknown id. */
int dqp_dyn_sizes(int id, int32_t *n_state, int32_t *n_ctrl);

/*
 * x_next = f(x, u) for n samples: x (n,n_state), u (n,n_ctrl), step dt.
 * Replaces: Dynamics.forward (deqmpc/my_envs/dynamics.py:26-63), PendulumDynamics.forward
 *           (deqmpc/envs.py:16-31), PendulumDx.forward (qpth/env_dx/pendulum.py:49-83).
 */
int dqp_dyn_step(int id, int32_t n, const double *x, const double *u, double dt, double *x_next,
                 void *stream);

/*
 * x_next and the Jacobians Jx (n,n_state,n_state) = d x_next_i / d x_j, Ju (n,n_state,n_ctrl).
 * Any output may be NULL.  Replaces: Dynamics.dynamics_derivatives / derivatives
 * (deqmpc/my_envs/dynamics.py:66-112,253-263), PendulumDynamics_jac (deqmpc/envs.py:68-82) -- the
 * `dx_jac(x, u) -> (x_next, (Jx, Ju))` closure the MPC layers call (qp_wrapper.py:497,
 * al_utils.py:212-262).
 */
int dqp_dyn_jacobian(int id, int32_t n, const double *x, const double *u, double dt, double *x_next,
                     double *Jx, double *Ju, void *stream);

/*
 * The reference extension's own interface (deqmpc/my_envs/cartpole1l/src/dynamics_cpu.cpp:8-56,
 * dynamics_gpu.cu kernels): q, qdot, tau (n,nq), per-sample step h (n); the six Jacobian blocks
 * are (n,nq,nq) with block[in i][out j], i.e. the raw CasADi buffers the reference's wrapper
 * concatenates and transposes (dynamics.py:99-112).  Robots 1-3 only.  Any block may be NULL.
 */
int dqp_dyn_forward_dynamics(int id, int32_t n, const double *q, const double *qdot, const double *tau,
                             const double *h, double *q_out, double *qdot_out, void *stream);
int dqp_dyn_forward_derivatives(int id, int32_t n, const double *q, const double *qdot,
                                const double *tau, const double *h, double *q_jac_q,
                                double *q_jac_qdot, double *q_jac_tau, double *qdot_jac_q,
                                double *qdot_jac_qdot, double *qdot_jac_tau, void *stream);

/* ------------------------------------------------------------------------------------------
 * Per-launch timing of the library's own kernels (bench.py's roofline figures, tools/).  Between
 * dqp_trace_begin and dqp_trace_end every kernel launch of this library is bracketed by a pair of HIP
 * events recorded on the launch stream; dqp_trace_end waits for them and returns one record per
 * launch, in launch order.  The one place the library owns anything: the event pool lives from begin to
 * end.  Not for use under stream capture.  Single host thread.
 */
typedef struct dqp_trace_record {
    char kernel[248];   /* demangled kernel name, truncated */
    float ms;           /* hipEventElapsedTime between the two events of this launch */
    int32_t reserved;
} dqp_trace_record;
int dqp_trace_begin(int32_t max_launches);      /* launches beyond max_launches are not recorded */
int dqp_trace_end(dqp_trace_record *out, int32_t capacity, int32_t *count);  /* out: HOST memory */

#ifdef __cplusplus
}
#endif
#endif /* DQP_H_ */
