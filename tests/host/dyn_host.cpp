// Host instantiation of the device dynamics templates (diff-qp-mpc_amd/csrc/dqp_dyn_models.h), for
// the CPU-side tests only: the same model code the HIP kernels inline, compiled for the host so
// that `pytest -m "not gpu"` can check the equations of motion against the reference's CasADi C
// (oracle/_ref) and the committed goldens without a GPU.  TEST INFRASTRUCTURE: never shipped in,
// or loaded by, the product library.
#include "../../diff-qp-mpc_amd/csrc/dqp_dyn_models.h"

using namespace dqp::dyn;

template <class Map> static void jac(int N, const double *x, const double *u, double dt, double *xn, double *Jx, double *Ju)
{
    constexpr int NX = Map::NX, NU = Map::NU, K = NX + NU;
    using S = Dual<K>;
    for (int i = 0; i < N; ++i) {
        S xs[NX], us[NU], out[NX];
        for (int k = 0; k < NX; ++k) { xs[k] = S(x[i * NX + k]); xs[k].d[k] = 1.0; }
        for (int k = 0; k < NU; ++k) { us[k] = S(u[i * NU + k]); us[k].d[NX + k] = 1.0; }
        Map::template step<S>(xs, us, dt, out);
        for (int r = 0; r < NX; ++r) {
            xn[i * NX + r] = out[r].v;
            for (int c = 0; c < NX; ++c) Jx[(i * NX + r) * NX + c] = out[r].d[c];
            for (int c = 0; c < NU; ++c) Ju[(i * NX + r) * NU + c] = out[r].d[NX + c];
        }
    }
}

extern "C" __attribute__((visibility("default"))) int dyn_host_jac(int id, int N, const double *x, const double *u,
                                                                   double dt, double *xn, double *Jx, double *Ju)
{
    switch (id) {
    case 1: jac<Robot<Pendulum1l>>(N, x, u, dt, xn, Jx, Ju); return 0;
    case 2: jac<Robot<Cartpole1l>>(N, x, u, dt, xn, Jx, Ju); return 0;
    case 3: jac<Robot<Cartpole2l>>(N, x, u, dt, xn, Jx, Ju); return 0;
    case 4: jac<PendulumEuler>(N, x, u, dt, xn, Jx, Ju); return 0;
    case 5: jac<PendulumDx>(N, x, u, dt, xn, Jx, Ju); return 0;
    case 6: jac<RexQuadrotor>(N, x, u, dt, xn, Jx, Ju); return 0;
    }
    return -1;
}
