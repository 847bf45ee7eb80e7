// dqp_trace.h -- every kernel launch of libdqp_hip.so goes through DQP_LAUNCH: with tracing on
// (dqp_trace_begin .. dqp_trace_end, include/dqp.h) the launch is bracketed by two HIP events recorded on its own
// stream; off, it is hipLaunchKernelGGL plus one load of a global.
#ifndef DQP_TRACE_H_
#define DQP_TRACE_H_
#include <hip/hip_runtime.h>

namespace dqp {
namespace trace {
extern int g_on;
void begin(const void *kernel, hipStream_t s);
void end(hipStream_t s);
}  // namespace trace
}  // namespace dqp

#define DQP_LAUNCH(KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                                          \
    do {                                                                                                          \
        if (dqp::trace::g_on) dqp::trace::begin(reinterpret_cast<const void *>(KERNEL), (hipStream_t)(STREAM));   \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, (hipStream_t)(STREAM), __VA_ARGS__);                         \
        if (dqp::trace::g_on) dqp::trace::end((hipStream_t)(STREAM));                                             \
    } while (0)

#endif
