// dqp_trace.hip -- dqp_trace_begin / dqp_trace_end (include/dqp.h): HIP-event timing of the library's own launches.
// Host code only.
#include <cxxabi.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/dqp.h"
#include "dqp_trace.h"

namespace dqp {
namespace trace {

int g_on = 0;

namespace {
struct Launch {
    const void *kernel;
    hipEvent_t a, b;
};
std::vector<Launch> g_launches;
std::vector<hipEvent_t> g_pool;
size_t g_next = 0;
bool g_open = false;       // an event pair was started and not closed yet
}  // namespace

void begin(const void *kernel, hipStream_t s)
{
    g_open = false;
    if (g_next + 2 > g_pool.size()) return;
    Launch l = {kernel, g_pool[g_next], g_pool[g_next + 1]};
    if (hipEventRecord(l.a, s) != hipSuccess) return;
    g_next += 2;
    g_launches.push_back(l);
    g_open = true;
}

void end(hipStream_t s)
{
    if (!g_open) return;
    g_open = false;
    (void)hipEventRecord(g_launches.back().b, s);
}

}  // namespace trace
}  // namespace dqp

using namespace dqp::trace;

extern "C" __attribute__((visibility("default"))) int dqp_trace_begin(int32_t max_launches)
{
    if (g_on || max_launches <= 0) return DQP_ERR_BAD_ARG;
    g_launches.clear();
    g_launches.reserve(max_launches);
    g_pool.assign(2 * (size_t)max_launches, nullptr);
    for (auto &e : g_pool)
        if (hipEventCreate(&e) != hipSuccess) {
            for (auto &d : g_pool) if (d) (void)hipEventDestroy(d);
            g_pool.clear();
            return DQP_ERR_LAUNCH;
        }
    g_next = 0;
    g_on = 1;
    return DQP_OK;
}

extern "C" __attribute__((visibility("default"))) int dqp_trace_end(dqp_trace_record *out, int32_t capacity, int32_t *count)
{
    if (!g_on) return DQP_ERR_BAD_ARG;
    g_on = 0;
    int n = 0, rc = DQP_OK;
    for (const Launch &l : g_launches) {
        float ms = 0.0f;
        if (hipEventSynchronize(l.b) != hipSuccess || hipEventElapsedTime(&ms, l.a, l.b) != hipSuccess) {
            rc = DQP_ERR_LAUNCH;
            break;
        }
        if (out && n < capacity) {
            const char *mangled = hipKernelNameRefByPtr(l.kernel, nullptr);
            int st = 1;
            char *dem = mangled ? abi::__cxa_demangle(mangled, nullptr, nullptr, &st) : nullptr;
            const char *name = (st == 0 && dem) ? dem : (mangled ? mangled : "?");
            strncpy(out[n].kernel, name, sizeof(out[n].kernel) - 1);
            out[n].kernel[sizeof(out[n].kernel) - 1] = 0;
            out[n].ms = ms;
            out[n].reserved = 0;
            free(dem);
        }
        ++n;
    }
    for (auto &e : g_pool) if (e) (void)hipEventDestroy(e);
    g_pool.clear();
    g_launches.clear();
    if (count) *count = n;
    return rc;
}
