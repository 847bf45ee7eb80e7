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
// Buffer layout (dqp_termination_bytes): hist (maxIter, B) x {resid, mu} doubles | accumulators:
// three u64 masks (improved, notbelow, notabove: one bit per iteration) | int32 header[TERM_HDR]
// (header[0] = I*, header[2] = block arrival counter) then int32 best[B] (the iteration of each problem's
// best iterate over all of pass 1: pass 2 takes a problem back iff best[qp] >= I*), then
// (16-byte aligned) the iterate snapshots [maxIter][B][snapDim] of the null-space kernels: every
// improving iterate of pass 1, so that pass 2 is an epilogue (r16n::finish_kernel), not a re-solve.
// Pass 1 zeroes accumulators + header itself (term_zero_acc), so a forward call is three launches.
#include "dqp_common.h"

namespace dqp {

namespace {

constexpr int MAXIT = TERM_MAXIT;
// The three batch-wide tests of the rule are ANDs / ORs of per-sample predicates, so each sample
// contributes one bit per iteration and the batch reduction is an OR of 64-bit masks:
struct Acc {
    unsigned long long improved;   // bit it: some sample improved at iteration it
    unsigned long long notbelow;   // bit it: some sample has NOT best_resid < eps (=> .max() < eps fails;
                                   //         a NaN best residual lands here, as torch's NaN max does)
    unsigned long long notabove;   // bit it: some sample has NOT mu > 1e32 (=> mu.min() > 1e32 fails)
};
static_assert(sizeof(Acc) + TERM_HDR * sizeof(int32_t) == TERM_ACC_BYTES, "layout shared with dqp_common.h");

inline size_t hist_bytes(int B, int maxIter) { return (size_t)B * maxIter * 2 * sizeof(double); }

__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
    return v;
}

// the reference's rule on the batch-wide masks -> the iteration count it stops at
__device__ __forceinline__ int rule_stop(unsigned long long ai, unsigned long long anb, unsigned long long ana, int maxIter,
                                         int notImprovedLim)
{
    int nNot = 0;
    for (int it = 0; it < maxIter; ++it) {
        if (it == 0 || ((ai >> it) & 1ull)) nNot = 0;
        else nNot += 1;
        if (nNot == notImprovedLim || !((anb >> it) & 1ull) || !((ana >> it) & 1ull)) return it + 1;
    }
    return maxIter;
}

// One thread per problem walks its history (best-so-far, the iteration it was found at: stored in
// hdr[TERM_HDR + qp], pass 2 compares it with I*) and collects its three masks; one OR-reduction per
// wavefront, one atomicOr per block.  The last block to finish replays the reference's rule on the masks
// and writes I* (hdr[0]; hdr[2] is the arrival counter).  The history is read SCAN_CHUNK iterations at a time
// (independent loads: the default max_iter = 20 is one round trip), and nothing here is O(B) on one thread: the first version's last block turned the
// best-iteration list into flags in a loop of dependent L2 round trips, two thirds of its 15 us.
constexpr int SCAN_CHUNK = 32;
__global__ __launch_bounds__(64) void term_scan_kernel(const double2 *hist, Acc *acc, int32_t *hdr,
                                                       int B, int maxIter, int notImprovedLim, double eps,
                                                       int decide)
{
    const int lane = threadIdx.x;
    const long long qp = (long long)blockIdx.x * blockDim.x + lane;
    const bool live = qp < B;
    unsigned long long imp = 0ull, nb = 0ull, na = 0ull;
    if (live) {
        const double2 *h = hist + qp;
        double best = 0.0;
        int arg = 0;
        for (int it0 = 0; it0 < maxIter; it0 += SCAN_CHUNK) {
            double2 v[SCAN_CHUNK];
#pragma unroll
            for (int k = 0; k < SCAN_CHUNK; ++k) v[k] = h[(long long)min(it0 + k, maxIter - 1) * B];
#pragma unroll
            for (int k = 0; k < SCAN_CHUNK; ++k) {
                const int it = it0 + k;
                if (it < maxIter) {
                    if (it == 0) best = v[k].x;                                   // batch.py:120-126
                    else if (v[k].x < best) { best = v[k].x; arg = it; imp |= 1ull << it; }
                    if (!(best < eps)) nb |= 1ull << it;
                    if (!(v[k].y > 1e32)) na |= 1ull << it;
                }
            }
        }
        hdr[TERM_HDR + qp] = arg;
    }
    imp = wave_or_u64(imp); nb = wave_or_u64(nb); na = wave_or_u64(na);
    if (lane < 3) {
        const unsigned long long m = lane == 0 ? imp : (lane == 1 ? nb : na);
        unsigned long long *dst = lane == 0 ? &acc->improved : (lane == 1 ? &acc->notbelow : &acc->notabove);
        if (m) atomicOr(dst, m);
    }
    if (!decide) return;            // multi-device form: the masks are combined across devices first
    // ---- last block: the rule itself
    __threadfence();
    int last = 0;
    if (lane == 0) last = (atomicAdd(&hdr[2], 1) == (int)gridDim.x - 1);
    last = __shfl(last, 0, 64);
    if (!last) return;
    __threadfence();
    if (lane == 0) {
        const unsigned long long ai = __hip_atomic_load(&acc->improved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long anb = __hip_atomic_load(&acc->notbelow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ana = __hip_atomic_load(&acc->notabove, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hdr[0] = rule_stop(ai, anb, ana, maxIter, notImprovedLim);
    }
}

// The rule on given masks (multi-device form: the OR of every shard's masks)
__global__ __launch_bounds__(64) void term_decide_kernel(const unsigned long long *masks, int32_t *hdr, int maxIter,
                                                         int notImprovedLim)
{
    if (threadIdx.x == 0) hdr[0] = rule_stop(masks[0], masks[1], masks[2], maxIter, notImprovedLim);
}

inline Acc *acc_of(void *term, int B, int maxIter) { return (Acc *)((char *)term + hist_bytes(B, maxIter)); }
inline int32_t *hdr_of(void *term, int B, int maxIter) { return (int32_t *)((char *)acc_of(term, B, maxIter) + sizeof(Acc)); }

inline size_t snap_offset(int B, int maxIter)
{
    const size_t o = hist_bytes(B, maxIter) + sizeof(Acc) + (size_t)(TERM_HDR + B) * sizeof(int32_t);
    return (o + 15) & ~(size_t)15;
}

}  // namespace

size_t term_bytes(int B, int maxIter, int snapDim)
{
    if (B <= 0 || maxIter <= 0) return 0;
    return snap_offset(B, maxIter) + (size_t)B * maxIter * (snapDim > 0 ? snapDim : 0) * sizeof(double);
}

void term_bind_pass1(KParams &P, void *term, int snapDim)
{
    P.hist = (double *)term;
    P.histIters = P.maxIter;
    P.cap = nullptr;
    P.histIn = nullptr;
    P.snap = snapDim > 0 ? (double *)((char *)term + snap_offset(P.B, P.maxIter)) : nullptr;
}

void term_bind_pass2(KParams &P, void *term)
{
    P.hist = nullptr;
    P.histIn = (const double *)term;
    P.cap = hdr_of(term, P.B, P.maxIter);         // P.snap stays as pass 1 had it
}

int term_decide(const KParams &P, void *term, void *stream)
{
    const int blocks = (P.B + 63) / 64;
    DQP_LAUNCH(term_scan_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream,
                       (const double2 *)term, acc_of(term, P.B, P.maxIter), hdr_of(term, P.B, P.maxIter),
                       P.B, P.maxIter, P.notImprovedLim, P.eps, 1);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

// multi-device form, step 1: this shard's three iteration masks -> masks[3] (device memory)
int term_local_masks(const KParams &P, void *term, unsigned long long *masks, void *stream)
{
    const int blocks = (P.B + 63) / 64;
    DQP_LAUNCH(term_scan_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream,
                       (const double2 *)term, acc_of(term, P.B, P.maxIter), hdr_of(term, P.B, P.maxIter),
                       P.B, P.maxIter, P.notImprovedLim, P.eps, 0);
    if (hipMemcpyAsync(masks, acc_of(term, P.B, P.maxIter), 3 * sizeof(unsigned long long), hipMemcpyDeviceToDevice,
                       (hipStream_t)stream) != hipSuccess)
        return DQP_ERR_LAUNCH;
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

// multi-device form, step 2: the rule on the combined masks -> I* of this shard's header
int term_decide_global(const KParams &P, void *term, const unsigned long long *masks, void *stream)
{
    DQP_LAUNCH(term_decide_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, masks,
                       hdr_of(term, P.B, P.maxIter), P.maxIter, P.notImprovedLim);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // namespace dqp
