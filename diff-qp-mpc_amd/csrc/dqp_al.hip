// dqp_al.hip -- Newton step of the augmented-Lagrangian MPC solver.
//
// Replaces, per Newton step of qpth/al_utils.py NewtonAL.forward (al_utils.py:403-427):
//   merit_hess = diag(Q) + rho * Jc^T Jc            (al_utils.py:96-102,176-178: a bmm per step)
//   U, info    = torch.linalg.cholesky_ex(merit_hess)
//   update     = -torch.cholesky_solve(grad, U)
// and, in NewtonAL.backward (al_utils.py:465-482),  inp_grad = -cholesky_solve(x_grad, U).
//
// One problem per 256-thread workgroup (4 wavefronts); the nz x nz Hessian lives in LDS
// (nz <= 128 -> <= 132 KB).  Jc^T Jc is the one real GEMM on this path (M = N = nz, K = ncon,
// e.g. 100 x 100 x 120 for cartpole T=20), so it runs on the fp64 matrix cores:
// v_mfma_f64_16x16x4_f64 tiles, lower-triangular tile pairs dealt round-robin to the 4 waves;
// a lane's A and B operands are both plain coalesced reads of one row segment of Jc.
// The Cholesky factor is right-looking in LDS on a 16x16 thread grid, the two triangular
// solves run in place, and L is written out (zero upper part, like torch) for backward.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/dqp.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

struct AlP {
    const double *Jc, *Qd, *rho, *grad, *Lin, *rhs;
    double *update, *L, *out;
    int32_t *info;
    int B, nz, ncon, ld;
};

// forward / backward substitution with the lower factor in LDS; b in LDS; all threads call.
__device__ __forceinline__ void chol_solve_lds(const double *H, int ld, int nz, double *b, int tid)
{
    for (int k = 0; k < nz; ++k) {                       // L y = b
        __syncthreads();
        const double yk = b[k] / H[k * ld + k];
        __syncthreads();
        if (tid == 0) b[k] = yk;
        for (int i = k + 1 + tid; i < nz; i += 256) b[i] = fma(-H[i * ld + k], yk, b[i]);
    }
    for (int k = nz - 1; k >= 0; --k) {                  // L^T x = y
        __syncthreads();
        const double xk = b[k] / H[k * ld + k];
        __syncthreads();
        if (tid == 0) b[k] = xk;
        for (int i = tid; i < k; i += 256) b[i] = fma(-H[k * ld + i], xk, b[i]);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void al_newton_kernel(AlP P)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nz = P.nz, ncon = P.ncon, ld = P.ld;
    const long long prob = blockIdx.x;
    double *H = sm;                    // nzp x ld
    double *bvec = sm + (size_t)((nz + 15) & ~15) * ld;
    // failure flag in the dynamic region (a static __shared__ would shift its base off 16 B)
    volatile double *s_info = bvec + ((nz + 15) & ~15);
    if (tid == 0) *s_info = 0.0;

    // ---- H = diag(Qd) + rho * Jc^T Jc with fp64 MFMA tiles ---------------------------------
    const double *J = P.Jc + prob * (long long)ncon * nz;
    const double rho = P.rho[prob];
    const int nt = (nz + 15) >> 4;
    int tile = 0;
    for (int ti = 0; ti < nt; ++ti)
        for (int tj = 0; tj <= ti; ++tj, ++tile) {
            if ((tile & 3) != wave) continue;
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
            const int ia = 16 * ti + (lane & 15), jb = 16 * tj + (lane & 15), kq = lane >> 4;
            for (int c0 = 0; c0 < ncon; c0 += 4) {
                const int c = c0 + kq;
                const bool cv = c < ncon;
                const double a = (cv && ia < nz) ? J[(long long)c * nz + ia] : 0.0;   // A[i][k] = Jc[k][i]
                const double b = (cv && jb < nz) ? J[(long long)c * nz + jb] : 0.0;   // B[k][j] = Jc[k][j]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
            // C/D layout of the f64 16x16x4 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row = 16 * ti + (lane >> 4) + 4 * rg, col = 16 * tj + (lane & 15);
                if (row < nz && col < nz) {
                    double v = rho * acc[rg];
                    if (row == col) v += P.Qd[prob * nz + row];
                    H[row * ld + col] = v;
                }
            }
        }
    for (int i = tid; i < nz; i += 256) bvec[i] = -P.grad[prob * nz + i];
    __syncthreads();

    // ---- right-looking Cholesky (lower) on a 16 x 16 thread grid ----------------------------
    const int ty = tid >> 4, tx = tid & 15;
    for (int k = 0; k < nz; ++k) {
        const double d = H[k * ld + k];
        __syncthreads();
        if (!(d > 0.0)) {                      // torch.linalg.cholesky_ex: info = first bad minor
            if (tid == 0 && *s_info == 0.0) *s_info = (double)(k + 1);
            break;
        }
        const double sq = sqrt(d), r = 1.0 / sq;
        if (tid == 0) H[k * ld + k] = sq;
        for (int i = k + 1 + tid; i < nz; i += 256) H[i * ld + k] *= r;
        __syncthreads();
        for (int i = k + 1 + ty; i < nz; i += 16) {
            const double lik = H[i * ld + k];
            for (int j = k + 1 + tx; j <= i; j += 16) H[i * ld + j] = fma(-lik, H[j * ld + k], H[i * ld + j]);
        }
        __syncthreads();
    }
    __syncthreads();
    const int info = (int)*s_info;
    if (info == 0) chol_solve_lds(H, ld, nz, bvec, tid);

    // ---- outputs ------------------------------------------------------------------------------
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);
    for (int i = tid; i < nz; i += 256) P.update[prob * nz + i] = info == 0 ? bvec[i] : nanv;
    if (P.L) {
        double *Lo = P.L + prob * (long long)nz * nz;
        for (int e = tid; e < nz * nz; e += 256) {
            const int i = e / nz, j = e - i * nz;
            Lo[e] = j <= i ? H[i * ld + j] : 0.0;
        }
    }
    if (tid == 0 && P.info) P.info[prob] = info;
}

// out = -(L L^T)^-1 rhs   (NewtonAL.backward, al_utils.py:477-480)
__global__ __launch_bounds__(256) void al_chol_solve_kernel(AlP P)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x;
    const int nz = P.nz, ld = P.ld;
    const long long prob = blockIdx.x;
    double *H = sm;
    double *bvec = sm + (size_t)((nz + 15) & ~15) * ld;
    const double *Li = P.Lin + prob * (long long)nz * nz;
    for (int e = tid; e < nz * nz; e += 256) {
        const int i = e / nz, j = e - i * nz;
        H[i * ld + j] = Li[e];
    }
    for (int i = tid; i < nz; i += 256) bvec[i] = -P.rhs[prob * nz + i];
    __syncthreads();
    chol_solve_lds(H, ld, nz, bvec, tid);
    for (int i = tid; i < nz; i += 256) P.out[prob * nz + i] = bvec[i];
}

int fill(const dqp_al_dims *d, AlP &P, size_t &lds)
{
    if (!d || d->nbatch < 0 || d->nz <= 0 || d->ncon < 0) return DQP_ERR_BAD_ARG;
    if (d->nz > 128) return DQP_ERR_TOO_LARGE;
    P.B = d->nbatch; P.nz = d->nz; P.ncon = d->ncon;
    P.ld = d->nz | 1;
    lds = ((size_t)((d->nz + 15) & ~15) * P.ld + ((d->nz + 15) & ~15) + 2) * sizeof(double);
    if (lds > 160 * 1024 - 64) return DQP_ERR_TOO_LARGE;
    return DQP_OK;
}

template <typename K>
int launch(K kernel, const AlP &P, size_t lds, void *stream)
{
    if (P.B == 0) return DQP_OK;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DQP_ERR_LAUNCH;
    hipLaunchKernelGGL(kernel, dim3(P.B), dim3(256), lds, (hipStream_t)stream, P);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int
dqp_al_newton_step(const dqp_al_dims *dims, const double *Jc, const double *Qdiag, const double *rho,
                   const double *grad, double *update, double *L, int32_t *info, void *stream)
{
    AlP P = {};
    size_t lds = 0;
    int rc = fill(dims, P, lds);
    if (rc) return rc;
    if (P.B == 0) return DQP_OK;
    if ((!Jc && P.ncon > 0) || !Qdiag || !rho || !grad || !update) return DQP_ERR_BAD_ARG;
    P.Jc = Jc; P.Qd = Qdiag; P.rho = rho; P.grad = grad; P.update = update; P.L = L; P.info = info;
    return launch(al_newton_kernel, P, lds, stream);
}

__attribute__((visibility("default"))) int
dqp_al_chol_solve(const dqp_al_dims *dims, const double *L, const double *rhs, double *out, void *stream)
{
    AlP P = {};
    size_t lds = 0;
    int rc = fill(dims, P, lds);
    if (rc) return rc;
    if (P.B == 0) return DQP_OK;
    if (!L || !rhs || !out) return DQP_ERR_BAD_ARG;
    P.Lin = L; P.rhs = rhs; P.out = out;
    return launch(al_chol_solve_kernel, P, lds, stream);
}

}  // extern "C"
