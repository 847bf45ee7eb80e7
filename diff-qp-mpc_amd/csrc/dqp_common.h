// dqp_common.h -- kernel parameter block shared by the HIP translation units of libdqp_hip.so
#ifndef DQP_COMMON_H_
#define DQP_COMMON_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dqp.h"

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
};


// DPP-row kernels (dqp_r16.hip): 4 QPs per wavefront for compile-time sizes <= 32.
// Return DQP_OK / error, or 1 when no instantiation matches (caller falls back to the
// generic kernels of dqp_pdipm.hip).
extern unsigned long long *g_debug_stamps;
int r16_forward(const KParams &P, void *stream);
int r16_backward(const KParams &P, void *stream);
// null-space form of the forward kernel (dqp_r16n.hip); same return convention.  It parks
// r16n_workspace_doubles(N, M, E) doubles per QP in P.workspace (0: no instantiation).
int r16n_forward(const KParams &P, void *stream);
long long r16n_workspace_doubles(int N, int M, int E);
// backward restarted from the context r16n_forward left in P.workspace (DQP_FLAG_BACKWARD_CTX)
int r16n_backward(const KParams &P, void *stream);

}  // namespace dqp
#endif
