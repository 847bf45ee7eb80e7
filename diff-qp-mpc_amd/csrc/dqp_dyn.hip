// dqp_dyn.hip -- device dynamics registry behind the C ABI (include/dqp.h, dqp_dyn_*).
//
// Replaces the reference's per-robot torch extensions (deqmpc/my_envs/{pendulum1l,cartpole1l,
// cartpole2l}: dynamics(q, qdot, tau, h) / derivatives(q, qdot, tau, h), CPU twin
// cartpole1l/src/dynamics_cpu.cpp:8-56, CUDA kernels src/dynamics_gpu.cu) and evaluates its two
// torch pendulum modules (deqmpc/envs.py:5-82, qpth/env_dx/pendulum.py:18-83) without Python.
// One thread per sample, everything in registers; the models live in dqp_dyn_models.h so that the
// MPC / AL kernels can inline the same code.  Jacobians are forward-mode duals of the same
// templates, a few seed directions per pass.
//
// HBM traffic is the algorithmic minimum: a step reads (n + m) and writes n doubles per sample;
// the Jacobian kernel additionally writes n (n + m).
#include "dqp_common.h"
#include "dqp_dyn_models.h"

namespace dqp {
namespace dyn {
namespace {

template <class Map>
__global__ __launch_bounds__(256) void step_kernel(int N, const double *x, const double *u, double dt, double *xn)
{
    constexpr int NX = Map::NX, NU = Map::NU;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        double xs[NX], us[NU], out[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) xs[k] = x[i * NX + k];
#pragma unroll
        for (int k = 0; k < NU; ++k) us[k] = u[i * NU + k];
        Map::template step<double>(xs, us, dt, out);
#pragma unroll
        for (int k = 0; k < NX; ++k) xn[i * NX + k] = out[k];
    }
}

// x_next and Jx (N,NX,NX) = d x_next_i / d x_j, Ju (N,NX,NU); seeds in chunks of KC directions
template <class Map, int KC>
__global__ __launch_bounds__(256) void jac_kernel(int N, const double *x, const double *u, double dt, double *xn,
                                                  double *Jx, double *Ju)
{
    constexpr int NX = Map::NX, NU = Map::NU, NS = NX + NU;
    using S = Dual<KC>;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        double xv[NX], uv[NU];
#pragma unroll
        for (int k = 0; k < NX; ++k) xv[k] = x[i * NX + k];
#pragma unroll
        for (int k = 0; k < NU; ++k) uv[k] = u[i * NU + k];
#pragma unroll
        for (int c0 = 0; c0 < NS; c0 += KC) {
            // the passes are independent: without this the scheduler interleaves them and the register
            // demand is that of all passes together (the 12-state model: 3 KB of scratch per lane)
#pragma unroll
            for (int k = 0; k < NX; ++k) asm volatile("" : "+v"(xv[k]) : : "memory");
            S xs[NX], us[NU], out[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                xs[k] = S(xv[k]);
                if (k >= c0 && k < c0 + KC) xs[k].d[k - c0] = 1.0;
            }
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                us[k] = S(uv[k]);
                if (NX + k >= c0 && NX + k < c0 + KC) us[k].d[NX + k - c0] = 1.0;
            }
            Map::template step<S>(xs, us, dt, out);
            if (c0 == 0 && xn) {
#pragma unroll
                for (int k = 0; k < NX; ++k) xn[i * NX + k] = out[k].v;
            }
#pragma unroll
            for (int r = 0; r < NX; ++r) {
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const int col = c0 + c;
                    if (col < NX) { if (Jx) Jx[(i * NX + r) * NX + col] = out[r].d[c]; }
                    else if (col < NS) { if (Ju) Ju[(i * NX + r) * NU + (col - NX)] = out[r].d[c]; }
                }
            }
        }
    }
}

// the reference extension's interface: per-sample step h, torque on every joint
template <class Model>
__global__ __launch_bounds__(256) void fd_kernel(int N, const double *q, const double *qd, const double *tau,
                                                 const double *h, double *qn, double *qdn)
{
    constexpr int NQ = Model::NQ;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        double a[NQ], b[NQ], t[NQ], o1[NQ], o2[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) { a[k] = q[i * NQ + k]; b[k] = qd[i * NQ + k]; t[k] = tau[i * NQ + k]; }
        rk4_step<Model, double>(a, b, t, h[i], o1, o2);
#pragma unroll
        for (int k = 0; k < NQ; ++k) { qn[i * NQ + k] = o1[k]; qdn[i * NQ + k] = o2[k]; }
    }
}

// six blocks (N,NQ,NQ), block[in i][out j] (the raw CasADi buffers the reference returns):
// q_jac_q, q_jac_qdot, q_jac_tau, qdot_jac_q, qdot_jac_qdot, qdot_jac_tau
template <class Model>
__global__ __launch_bounds__(256) void fdd_kernel(int N, const double *q, const double *qd, const double *tau,
                                                  const double *h, double *b0, double *b1, double *b2,
                                                  double *b3, double *b4, double *b5)
{
    constexpr int NQ = Model::NQ;
    using S = Dual<NQ>;
    double *qblk[3] = {b0, b1, b2}, *vblk[3] = {b3, b4, b5};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {           // seed group: q, qdot, tau
            S a[NQ], b[NQ], t[NQ], o1[NQ], o2[NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                a[k] = S(q[i * NQ + k]); b[k] = S(qd[i * NQ + k]); t[k] = S(tau[i * NQ + k]);
                if (g == 0) a[k].d[k] = 1.0;
                if (g == 1) b[k].d[k] = 1.0;
                if (g == 2) t[k].d[k] = 1.0;
            }
            rk4_step<Model, S>(a, b, t, S(h[i]), o1, o2);
#pragma unroll
            for (int in = 0; in < NQ; ++in) {
#pragma unroll
                for (int out = 0; out < NQ; ++out) {
                    if (qblk[g]) qblk[g][(i * NQ + in) * NQ + out] = o1[out].d[in];
                    if (vblk[g]) vblk[g][(i * NQ + in) * NQ + out] = o2[out].d[in];
                }
            }
        }
    }
}

inline int grid_for(int N) { const int b = (N + 255) / 256; return b < 4096 ? b : 4096; }

template <class Map> int run_step(int N, const double *x, const double *u, double dt, double *xn, void *s)
{
    DQP_LAUNCH(step_kernel<Map>, dim3(grid_for(N)), dim3(256), 0, (hipStream_t)s, N, x, u, dt, xn);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}
template <class Map, int KC>
int run_jac(int N, const double *x, const double *u, double dt, double *xn, double *Jx, double *Ju, void *s)
{
    DQP_LAUNCH((jac_kernel<Map, KC>), dim3(grid_for(N)), dim3(256), 0, (hipStream_t)s, N, x, u, dt, xn, Jx, Ju);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}
template <class Model>
int run_fd(int N, const double *q, const double *qd, const double *tau, const double *h, double *qn, double *qdn, void *s)
{
    DQP_LAUNCH(fd_kernel<Model>, dim3(grid_for(N)), dim3(256), 0, (hipStream_t)s, N, q, qd, tau, h, qn, qdn);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}
template <class Model>
int run_fdd(int N, const double *q, const double *qd, const double *tau, const double *h, double *const *b, void *s)
{
    DQP_LAUNCH(fdd_kernel<Model>, dim3(grid_for(N)), dim3(256), 0, (hipStream_t)s, N, q, qd, tau, h,
                       b[0], b[1], b[2], b[3], b[4], b[5]);
    return hipGetLastError() == hipSuccess ? DQP_OK : DQP_ERR_LAUNCH;
}

}  // namespace
}  // namespace dyn
}  // namespace dqp

using namespace dqp::dyn;

extern "C" {

__attribute__((visibility("default"))) int dqp_dyn_sizes(int id, int32_t *n_state, int32_t *n_ctrl)
{
    int n, m = 1;
    switch (id) {
    case DQP_DYN_PENDULUM1L: n = 2; break;
    case DQP_DYN_CARTPOLE1L: n = 4; break;
    case DQP_DYN_CARTPOLE2L: n = 6; break;
    case DQP_DYN_PENDULUM_EULER: n = 2; break;
    case DQP_DYN_PENDULUM_DX: n = 3; break;
    case DQP_DYN_REXQUADROTOR: n = 12; m = 4; break;
    default: return DQP_ERR_BAD_ARG;
    }
    if (n_state) *n_state = n;
    if (n_ctrl) *n_ctrl = m;
    return DQP_OK;
}

__attribute__((visibility("default"))) int dqp_dyn_step(int id, int32_t n, const double *x, const double *u,
                                                        double dt, double *x_next, void *stream)
{
    if (n < 0) return DQP_ERR_BAD_ARG;
    if (dqp_dyn_sizes(id, nullptr, nullptr) != DQP_OK) return DQP_ERR_BAD_ARG;
    if (n == 0) return DQP_OK;
    if (!x || !u || !x_next) return DQP_ERR_BAD_ARG;
    switch (id) {
    case DQP_DYN_PENDULUM1L: return run_step<Robot<Pendulum1l>>(n, x, u, dt, x_next, stream);
    case DQP_DYN_CARTPOLE1L: return run_step<Robot<Cartpole1l>>(n, x, u, dt, x_next, stream);
    case DQP_DYN_CARTPOLE2L: return run_step<Robot<Cartpole2l>>(n, x, u, dt, x_next, stream);
    case DQP_DYN_PENDULUM_EULER: return run_step<PendulumEuler>(n, x, u, dt, x_next, stream);
    case DQP_DYN_REXQUADROTOR: return run_step<RexQuadrotor>(n, x, u, dt, x_next, stream);
    default: return run_step<PendulumDx>(n, x, u, dt, x_next, stream);
    }
}

__attribute__((visibility("default"))) int dqp_dyn_jacobian(int id, int32_t n, const double *x, const double *u,
                                                            double dt, double *x_next, double *Jx, double *Ju,
                                                            void *stream)
{
    if (n < 0) return DQP_ERR_BAD_ARG;
    if (dqp_dyn_sizes(id, nullptr, nullptr) != DQP_OK) return DQP_ERR_BAD_ARG;
    if (n == 0) return DQP_OK;
    if (!x || !u) return DQP_ERR_BAD_ARG;
    switch (id) {
    case DQP_DYN_PENDULUM1L: return run_jac<Robot<Pendulum1l>, 3>(n, x, u, dt, x_next, Jx, Ju, stream);
    case DQP_DYN_CARTPOLE1L: return run_jac<Robot<Cartpole1l>, 5>(n, x, u, dt, x_next, Jx, Ju, stream);
    case DQP_DYN_CARTPOLE2L: return run_jac<Robot<Cartpole2l>, 4>(n, x, u, dt, x_next, Jx, Ju, stream);
    case DQP_DYN_PENDULUM_EULER: return run_jac<PendulumEuler, 3>(n, x, u, dt, x_next, Jx, Ju, stream);
    case DQP_DYN_REXQUADROTOR: return run_jac<RexQuadrotor, 2>(n, x, u, dt, x_next, Jx, Ju, stream);
    default: return run_jac<PendulumDx, 4>(n, x, u, dt, x_next, Jx, Ju, stream);
    }
}

__attribute__((visibility("default"))) int dqp_dyn_forward_dynamics(int id, int32_t n, const double *q,
                                                                    const double *qdot, const double *tau,
                                                                    const double *h, double *q_out,
                                                                    double *qdot_out, void *stream)
{
    if (n < 0) return DQP_ERR_BAD_ARG;
    if (id != DQP_DYN_PENDULUM1L && id != DQP_DYN_CARTPOLE1L && id != DQP_DYN_CARTPOLE2L) return DQP_ERR_BAD_ARG;
    if (n == 0) return DQP_OK;
    if (!q || !qdot || !tau || !h || !q_out || !qdot_out) return DQP_ERR_BAD_ARG;
    switch (id) {
    case DQP_DYN_PENDULUM1L: return run_fd<Pendulum1l>(n, q, qdot, tau, h, q_out, qdot_out, stream);
    case DQP_DYN_CARTPOLE1L: return run_fd<Cartpole1l>(n, q, qdot, tau, h, q_out, qdot_out, stream);
    default: return run_fd<Cartpole2l>(n, q, qdot, tau, h, q_out, qdot_out, stream);
    }
}

__attribute__((visibility("default"))) int
dqp_dyn_forward_derivatives(int id, int32_t n, const double *q, const double *qdot, const double *tau,
                            const double *h, double *q_jac_q, double *q_jac_qdot, double *q_jac_tau,
                            double *qdot_jac_q, double *qdot_jac_qdot, double *qdot_jac_tau, void *stream)
{
    if (n < 0) return DQP_ERR_BAD_ARG;
    if (id != DQP_DYN_PENDULUM1L && id != DQP_DYN_CARTPOLE1L && id != DQP_DYN_CARTPOLE2L) return DQP_ERR_BAD_ARG;
    if (n == 0) return DQP_OK;
    if (!q || !qdot || !tau || !h) return DQP_ERR_BAD_ARG;
    double *const b[6] = {q_jac_q, q_jac_qdot, q_jac_tau, qdot_jac_q, qdot_jac_qdot, qdot_jac_tau};
    switch (id) {
    case DQP_DYN_PENDULUM1L: return run_fdd<Pendulum1l>(n, q, qdot, tau, h, b, stream);
    case DQP_DYN_CARTPOLE1L: return run_fdd<Cartpole1l>(n, q, qdot, tau, h, b, stream);
    default: return run_fdd<Cartpole2l>(n, q, qdot, tau, h, b, stream);
    }
}

}  // extern "C"
