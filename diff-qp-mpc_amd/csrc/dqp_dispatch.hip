// dqp_dispatch.hip -- size table of the DPP-row kernels.  _build.py passes the lists as
//   -DDQP_R16_SIZE_LIST="X(30,30,15) X(20,10,15) ..."  and  -DDQP_R16N_SIZE_LIST="..."
// and compiles one object per entry from dqp_r16.hip / dqp_r16n.hip.
#include "dqp_common.h"

#ifndef DQP_R16_SIZE_LIST
#error "DQP_R16_SIZE_LIST not defined (see _build.py)"
#endif
#ifndef DQP_R16N_SIZE_LIST
#define DQP_R16N_SIZE_LIST
#endif

namespace dqp {

#define X(n, m, e)                                                  \
    int r16_forward_##n##_##m##_##e(const KParams &, void *);       \
    int r16_backward_##n##_##m##_##e(const KParams &, void *);
DQP_R16_SIZE_LIST
#undef X
#define X(n, m, e)                                                  \
    int r16n_forward_##n##_##m##_##e(const KParams &, void *);      \
    int r16n_backward_##n##_##m##_##e(const KParams &, void *);
DQP_R16N_SIZE_LIST
#undef X

int r16_forward(const KParams &P, void *stream)
{
#define X(n, m, e) if (P.N == n && P.M == m && P.E == e) return r16_forward_##n##_##m##_##e(P, stream);
    DQP_R16_SIZE_LIST
#undef X
    return 1;
}

int r16_backward(const KParams &P, void *stream)
{
#define X(n, m, e) if (P.N == n && P.M == m && P.E == e) return r16_backward_##n##_##m##_##e(P, stream);
    DQP_R16_SIZE_LIST
#undef X
    return 1;
}

int r16n_forward(const KParams &P, void *stream)
{
#define X(n, m, e) if (P.N == n && P.M == m && P.E == e) return r16n_forward_##n##_##m##_##e(P, stream);
    DQP_R16N_SIZE_LIST
#undef X
    return 1;
}

int r16n_backward(const KParams &P, void *stream)
{
#define X(n, m, e) if (P.N == n && P.M == m && P.E == e) return r16n_backward_##n##_##m##_##e(P, stream);
    DQP_R16N_SIZE_LIST
#undef X
    return 1;
}

long long r16n_workspace_doubles(int N, int M, int E)
{
#define X(n, m, e)                                                                          \
    if (N == n && M == m && E == e)                                                         \
        return (long long)e * (n - e) + (long long)e * (e - 1) / 2 + (long long)n * (n + 1) / 2 + \
               (long long)m * n + (long long)e * e + 5 * e + n + 1;   /* = r16n::Cfg::wsQP */
    DQP_R16N_SIZE_LIST
#undef X
    return 0;
}

// doubles per (problem, iteration) of the iterate snapshots the null-space kernels keep for the batch
// rule's finish pass (= r16n::Cfg::snapDim); 0 when the size has no null-space kernel
int r16n_snapshot_doubles(int N, int M, int E)
{
    return r16n_workspace_doubles(N, M, E) > 0 ? (N - E) + 2 * M + 2 : 0;
}

}  // namespace dqp
