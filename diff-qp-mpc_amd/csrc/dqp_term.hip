// dqp_term.hip -- the reference's batch-coupled stopping rule, replayed on the device.
//
// Reference (qpth/solvers/pdipm/batch.py:119-144, identically batch_LU.py:118-147): per iteration
//     I = resids < best_resids            (per sample; the first iteration just stores resids)
//     nNotImproved = 0 if I.sum() > 0 else nNotImproved + 1
//     stop if nNotImproved == notImprovedLim or best_resids.max() < eps or mu.min() > 1e32
// and the value returned for a sample is its best iterate among the iterations executed.  Samples
// never interact otherwise, so the forward kernels iterate every problem to max_iter while
// recording (resid, mu) per iteration (pass 1); here a reduction over the batch finds the
// iteration count I* at which the reference stops, and marks the problems whose best iterate came
// at an iteration >= I* -- they are solved again with max_iter = I* (pass 2).  In a large batch
// some sample improves at every iteration, I* = max_iter and pass 2 is an empty launch.
//
// Buffer layout (dqp_termination_bytes): hist (B, maxIter) x {resid, mu} doubles | accumulators:
// u64 bestmax[64], u64 mumin_c[64], u64 improved, u64 bestnan, u64 munan | int32 header[TERM_HDR]
// (header[0] = I*, header[1] = number of problems to redo) then int32 redo[B] (pass 1 stores the
// problem's best iteration there, term_decide_kernel turns it into the flag).
#include "dqp_common.h"

namespace dqp {

namespace {

constexpr int MAXIT = 64;
struct Acc {
    unsigned long long bestmax[MAXIT];   // max over the batch of best_resids after iteration it (bit pattern)
    unsigned long long mumin_c[MAXIT];   // ~bits of min over the batch of mu at iteration it
    unsigned long long improved;         // bit it: some sample improved at iteration it
    unsigned long long bestnan;          // bit it: some sample's best residual is NaN (=> .max() is NaN)
    unsigned long long munan;            // bit it: some sample's mu is NaN (=> .min() is NaN)
};

inline size_t hist_bytes(int B, int maxIter) { return (size_t)B * maxIter * 2 * sizeof(double); }

__global__ __launch_bounds__(256) void term_scan_kernel(const double2 *hist, Acc *acc, int32_t *argbest,
                                                        int B, int maxIter)
{
    __shared__ unsigned long long s_max[MAXIT], s_min[MAXIT], s_imp, s_bn, s_mn;
    const int tid = threadIdx.x;
    if (tid < MAXIT) { s_max[tid] = 0ull; s_min[tid] = 0ull; }
    if (tid == 0) { s_imp = 0ull; s_bn = 0ull; s_mn = 0ull; }
    __syncthreads();
    const long long qp = (long long)blockIdx.x * blockDim.x + tid;
    if (qp < B) {
        const double2 *h = hist + qp * maxIter;
        double best = 0.0;
        int arg = 0;
        unsigned long long imp = 0ull, bn = 0ull, mn = 0ull;
        for (int it = 0; it < maxIter; ++it) {
            const double2 v = h[it];
            if (it == 0) { best = v.x; }                           // batch.py:120-126
            else if (v.x < best) { best = v.x; arg = it; imp |= 1ull << it; }
            if (best != best) bn |= 1ull << it;
            else atomicMax(&s_max[it], (unsigned long long)__double_as_longlong(best));
            if (v.y != v.y) mn |= 1ull << it;
            else atomicMax(&s_min[it], ~(unsigned long long)__double_as_longlong(v.y));
        }
        argbest[qp] = arg;
        if (imp) atomicOr(&s_imp, imp);
        if (bn) atomicOr(&s_bn, bn);
        if (mn) atomicOr(&s_mn, mn);
    }
    __syncthreads();
    if (tid < maxIter) {
        atomicMax(&acc->bestmax[tid], s_max[tid]);
        atomicMax(&acc->mumin_c[tid], s_min[tid]);
    }
    if (tid == 0) {
        if (s_imp) atomicOr(&acc->improved, s_imp);
        if (s_bn) atomicOr(&acc->bestnan, s_bn);
        if (s_mn) atomicOr(&acc->munan, s_mn);
    }
}

__global__ __launch_bounds__(256) void term_decide_kernel(const Acc *acc, int32_t *hdr, int B, int maxIter,
                                                          int notImprovedLim, double eps)
{
    // every thread replays the (<= 64-step) batch rule; cheaper than a second launch
    int istop = maxIter, nNot = 0;
    const unsigned long long imp = acc->improved, bn = acc->bestnan, mn = acc->munan;
    for (int it = 0; it < maxIter; ++it) {
        if (it == 0 || ((imp >> it) & 1ull)) nNot = 0;
        else nNot += 1;
        const bool best_ok = !((bn >> it) & 1ull) && __longlong_as_double((long long)acc->bestmax[it]) < eps;
        const bool mu_ok = !((mn >> it) & 1ull) && __longlong_as_double((long long)~acc->mumin_c[it]) > 1e32;
        if (nNot == notImprovedLim || best_ok || mu_ok) { istop = it + 1; break; }
    }
    const long long qp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (qp < B) {
        const int redo = hdr[TERM_HDR + qp] >= istop ? 1 : 0;
        hdr[TERM_HDR + qp] = redo;
        if (redo) atomicAdd(&hdr[1], 1);
    }
    if (qp == 0) hdr[0] = istop;
}

inline Acc *acc_of(void *term, int B, int maxIter) { return (Acc *)((char *)term + hist_bytes(B, maxIter)); }
inline int32_t *hdr_of(void *term, int B, int maxIter) { return (int32_t *)((char *)acc_of(term, B, maxIter) + sizeof(Acc)); }

}  // namespace

size_t term_bytes(int B, int maxIter)
{
    if (B <= 0 || maxIter <= 0) return 0;
    return hist_bytes(B, maxIter) + sizeof(Acc) + (size_t)(TERM_HDR + B) * sizeof(int32_t);
}

int term_clear(const KParams &P, void *term, void *stream)
{
    // accumulators and header to zero (hist and the redo list are fully overwritten by pass 1 / scan)
    return hipMemsetAsync(acc_of(term, P.B, P.maxIter), 0, sizeof(Acc) + TERM_HDR * sizeof(int32_t),
                          (hipStream_t)stream) == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

void term_bind_pass1(KParams &P, void *term)
{
    P.hist = (double *)term;
    P.histIters = P.maxIter;
    P.cap = nullptr;
}

void term_bind_pass2(KParams &P, void *term)
{
    P.hist = nullptr;
    P.cap = hdr_of(term, P.B, P.maxIter);
}

int term_decide(const KParams &P, void *term, void *stream)
{
    const int blocks = (P.B + 255) / 256;
    Acc *acc = acc_of(term, P.B, P.maxIter);
    int32_t *hdr = hdr_of(term, P.B, P.maxIter);
    hipLaunchKernelGGL(term_scan_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const double2 *)term, acc, hdr + TERM_HDR, P.B, P.maxIter);
    hipLaunchKernelGGL(term_decide_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const Acc *)acc, hdr, P.B, P.maxIter, P.notImprovedLim, P.eps);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // namespace dqp
