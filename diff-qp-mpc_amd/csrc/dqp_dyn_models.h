// dqp_dyn_models.h -- device-side dynamics registry: the robots the reference ships as CasADi
// generated C / CUDA extensions (deqmpc/my_envs/{pendulum1l,cartpole1l,cartpole2l}/src) and its two
// torch pendulum modules, written from their equations of motion so that any kernel can evaluate
// x_{t+1} = f(x_t, u_t) and its Jacobians in registers.
//
// The reference's generated code is a straight-line CasADi expression of one RK4 step of a rigid
// body model.  The models below were identified from it (tests/test_dynamics_cpu.py and
// tests/test_gpu_dyn.py compare with the reference's compiled C: agreement 2e-15 on the states,
// 5e-16 on the Jacobians) and are written as manipulator equations
//        M(q) qdd + c(q, qd) = tau + G(q)
//   pendulum1l  (nq 1): 0.25 thdd = tau - 4.905 sin th
//   cartpole1l  (nq 2): M = [[11, -c1], [-c1, 2]],  c = [s1 thd^2, 0],  G = 9.81 [0, s1]
//   cartpole2l  (nq 3): M = [[12, -(2 c1 + c12), -c12], [., 5 + 2 c2, 2 + c2], [., ., 2]],
//                       c = [2 s1 w1^2 + s12 (w1 + w2)^2, -s2 w2 (2 w1 + w2), s2 w1^2],
//                       G = 9.81 [0, 2 s1 + s12, s12]
// (cart mass 10, unit link masses / COM offsets / inertias; th = 0 upright for the cartpoles, hanging
// for pendulum1l), integrated by the classic RK4 scheme on (q, qd) with step h, the torque held.
//   pendulum_euler: deqmpc/envs.py:5-47 (semi-implicit Euler, thdd = u + 10 sin th, dt given)
//   pendulum_dx:    qpth/env_dx/pendulum.py:49-83 (state (cos th, sin th, thd), g=10, m=l=1,
//                   u clamped to +-2, explicit Euler on thd then th)
//   rexquadrotor:   deqmpc/rex_quadrotor.py:51-129 (12 states, 4 motors, MRP attitude, RK4)
//
// Everything is templated on the scalar type so that the same code evaluates values (double) and
// forward-mode derivatives (Dual<K>: K directional derivatives ride along in registers).
#ifndef DQP_DYN_MODELS_H_
#define DQP_DYN_MODELS_H_
#include <hip/hip_runtime.h>
#include <math.h>

namespace dqp {
namespace dyn {

// ------------------------------------------------------------------ forward-mode scalar
template <int K> struct Dual {
    double v;
    double d[K];
    __host__ __device__ __forceinline__ Dual() {}
    __host__ __device__ __forceinline__ Dual(double x) : v(x)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) d[k] = 0.0;
    }
};
#define DQP_DUAL_BIN(op, VAL, DA, DB)                                                         \
    template <int K> __host__ __device__ __forceinline__ Dual<K> operator op(const Dual<K> &a, const Dual<K> &b) \
    {                                                                                         \
        Dual<K> r;                                                                            \
        r.v = VAL;                                                                            \
        _Pragma("unroll") for (int k = 0; k < K; ++k) r.d[k] = (DA) * a.d[k] + (DB) * b.d[k];  \
        return r;                                                                             \
    }
DQP_DUAL_BIN(+, a.v + b.v, 1.0, 1.0)
DQP_DUAL_BIN(-, a.v - b.v, 1.0, -1.0)
DQP_DUAL_BIN(*, a.v * b.v, b.v, a.v)
#undef DQP_DUAL_BIN
template <int K> __host__ __device__ __forceinline__ Dual<K> operator/(const Dual<K> &a, const Dual<K> &b)
{
    Dual<K> r;
    const double ib = 1.0 / b.v;
    r.v = a.v * ib;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) * ib;
    return r;
}
template <int K> __host__ __device__ __forceinline__ Dual<K> operator-(const Dual<K> &a)
{
    Dual<K> r;
    r.v = -a.v;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = -a.d[k];
    return r;
}
template <int K> __host__ __device__ __forceinline__ Dual<K> operator*(double s, const Dual<K> &a)
{
    Dual<K> r;
    r.v = s * a.v;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = s * a.d[k];
    return r;
}
template <int K> __host__ __device__ __forceinline__ Dual<K> operator*(const Dual<K> &a, double s) { return s * a; }
template <int K> __host__ __device__ __forceinline__ Dual<K> operator+(const Dual<K> &a, double s) { Dual<K> r = a; r.v += s; return r; }
template <int K> __host__ __device__ __forceinline__ Dual<K> operator+(double s, const Dual<K> &a) { return a + s; }
template <int K> __host__ __device__ __forceinline__ Dual<K> operator-(const Dual<K> &a, double s) { return a + (-s); }
template <int K> __host__ __device__ __forceinline__ Dual<K> operator-(double s, const Dual<K> &a) { return (-a) + s; }

// (a hand-written device sincos -- fma Cody-Waite reduction + minimax kernels, ~40 instructions -- was measured
// against the device library's here in round 3: the library's small-argument path executes ~53, the line search
// moved by 1 %; not kept)
__host__ __device__ __forceinline__ void sincos_(double x, double &s, double &c) { sincos(x, &s, &c); }
template <int K> __host__ __device__ __forceinline__ void sincos_(const Dual<K> &x, Dual<K> &s, Dual<K> &c)
{
    double sv, cv;
    sincos(x.v, &sv, &cv);
    s.v = sv; c.v = cv;
#pragma unroll
    for (int k = 0; k < K; ++k) { s.d[k] = cv * x.d[k]; c.d[k] = -sv * x.d[k]; }
}
__host__ __device__ __forceinline__ double atan2_(double y, double x) { return atan2(y, x); }
template <int K> __host__ __device__ __forceinline__ Dual<K> atan2_(const Dual<K> &y, const Dual<K> &x)
{
    Dual<K> r;
    r.v = atan2(y.v, x.v);
    const double in = 1.0 / (x.v * x.v + y.v * y.v);
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = (x.v * y.d[k] - y.v * x.d[k]) * in;
    return r;
}
// torch.clamp: zero derivative outside (lo, hi), one inside and AT the bounds (clamp_backward masks
// with lo <= x <= hi)
__host__ __device__ __forceinline__ double clamp_(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
template <int K> __host__ __device__ __forceinline__ Dual<K> clamp_(const Dual<K> &x, double lo, double hi)
{
    Dual<K> r;
    r.v = fmin(fmax(x.v, lo), hi);
    const double m = (x.v >= lo && x.v <= hi) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) r.d[k] = m * x.d[k];
    return r;
}

// ------------------------------------------------------------------ rigid-body models (accelerations)
struct Pendulum1l {
    static constexpr int NQ = 1;
    template <class S> __host__ __device__ static void accel(const S *q, const S *, const S *tau, S *a)
    {
        S s, c;
        sincos_(q[0], s, c);
        a[0] = 4.0 * tau[0] - 19.62 * s;
    }
};

struct Cartpole1l {
    static constexpr int NQ = 2;
    template <class S> __host__ __device__ static void accel(const S *q, const S *qd, const S *tau, S *a)
    {
        S s, c;
        sincos_(q[1], s, c);
        const S r0 = tau[0] - s * qd[1] * qd[1];
        const S r1 = tau[1] + 9.81 * s;
        const S det = 22.0 - c * c;                 // det [[11, -c], [-c, 2]]
        a[0] = (2.0 * r0 + c * r1) / det;
        a[1] = (c * r0 + 11.0 * r1) / det;
    }
};

struct Cartpole2l {
    static constexpr int NQ = 3;
    template <class S> __host__ __device__ static void accel(const S *q, const S *qd, const S *tau, S *a)
    {
        S s1, c1, s2, c2, s12, c12;
        sincos_(q[1], s1, c1);
        sincos_(q[2], s2, c2);
        // sin / cos of q1 + q2 by the addition theorem (six flops) instead of a third range reduction + two polynomials:
        // a third of this model's instructions; the result differs from sin(q1 + q2) by rounding only (<= 2 ulp)
        s12 = s1 * c2 + c1 * s2;
        c12 = c1 * c2 - s1 * s2;
        const S w12 = qd[1] + qd[2];
        const S r0 = tau[0] - (2.0 * s1 * qd[1] * qd[1] + s12 * w12 * w12);
        const S r1 = tau[1] + 9.81 * (2.0 * s1 + s12) + s2 * qd[2] * (2.0 * qd[1] + qd[2]);
        const S r2 = tau[2] + 9.81 * s12 - s2 * qd[1] * qd[1];
        // symmetric 3x3 solve by LDL^T (M is positive definite)
        const S m01 = -(2.0 * c1 + c12), m02 = -c12, m11 = 5.0 + 2.0 * c2, m12 = 2.0 + c2;
        const double m00 = 12.0, m22 = 2.0;
        const S l10 = m01 * (1.0 / m00), l20 = m02 * (1.0 / m00);
        const S d1 = m11 - l10 * m01;
        const S l21 = (m12 - l20 * m01) / d1;
        const S d2 = m22 - l20 * m02 - l21 * l21 * d1;
        const S y0 = r0, y1 = r1 - l10 * y0, y2 = r2 - l20 * y0 - l21 * y1;
        const S x2 = y2 / d2;
        const S x1 = y1 / d1 - l21 * x2;
        a[2] = x2;
        a[1] = x1;
        a[0] = y0 * (1.0 / m00) - l10 * x1 - l20 * x2;
    }
};

// One classic RK4 step of (q, qd)' = (qd, accel(q, qd, tau)) with the torque held over the step
// (what the CasADi expression of deqmpc/my_envs/*/src/generated_dynamics.c evaluates).
template <class Model, class S>
__host__ __device__ __forceinline__ void rk4_step(const S *q, const S *qd, const S *tau, const S &h, S *qn, S *qdn)
{
    constexpr int NQ = Model::NQ;
    S k1[NQ], k2[NQ], k3[NQ], k4[NQ], tq[NQ], tv[NQ];
    const S hh = 0.5 * h;
    Model::accel(q, qd, tau, k1);
#pragma unroll
    for (int i = 0; i < NQ; ++i) { tq[i] = q[i] + hh * qd[i]; tv[i] = qd[i] + hh * k1[i]; }
    Model::accel(tq, tv, tau, k2);
    S v2[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) { v2[i] = tv[i]; tq[i] = q[i] + hh * v2[i]; tv[i] = qd[i] + hh * k2[i]; }
    Model::accel(tq, tv, tau, k3);
    S v3[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) { v3[i] = tv[i]; tq[i] = q[i] + h * v3[i]; tv[i] = qd[i] + h * k3[i]; }
    Model::accel(tq, tv, tau, k4);
    const S h6 = h * (1.0 / 6.0);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        qn[i] = q[i] + h6 * (qd[i] + 2.0 * (v2[i] + v3[i]) + tv[i]);
        qdn[i] = qd[i] + h6 * (k1[i] + 2.0 * (k2[i] + k3[i]) + k4[i]);
    }
}

// ------------------------------------------------------------------ state-space maps x+ = f(x, u, dt)
// Robot<Model>: x = [q, qd], u drives joint 0 (deqmpc/my_envs/dynamics.py:26-63).
template <class Model> struct Robot {
    static constexpr int NX = 2 * Model::NQ, NU = 1;
    template <class S> __host__ __device__ static void step(const S *x, const S *u, double dt, S *xn)
    {
        constexpr int NQ = Model::NQ;
        S tau[NQ];
        tau[0] = u[0];
#pragma unroll
        for (int i = 1; i < NQ; ++i) tau[i] = S(0.0);
        rk4_step<Model, S>(x, x + NQ, tau, S(dt), xn, xn + NQ);
    }
};

struct PendulumEuler {            // deqmpc/envs.py:16-47
    static constexpr int NX = 2, NU = 1;
    template <class S> __host__ __device__ static void step(const S *x, const S *u, double dt, S *xn)
    {
        S s, c;
        sincos_(x[0], s, c);
        const S acc = u[0] + 10.0 * s;          // (u + m g l sin th) / (m l^2), m = l = 1, g = 10
        xn[1] = x[1] + dt * acc;
        xn[0] = x[0] + dt * xn[1];
    }
};

struct PendulumDx {               // qpth/env_dx/pendulum.py:49-83 (simple=True, default params)
    static constexpr int NX = 3, NU = 1;
    template <class S> __host__ __device__ static void step(const S *x, const S *u, double dt, S *xn)
    {
        const S uc = clamp_(u[0], -2.0, 2.0);
        const S th = atan2_(x[1], x[0]);
        const S ndth = x[2] + dt * (15.0 * x[1] + 3.0 * uc);   // -3g/(2l) * (-sin th) + 3u/(m l^2)
        const S nth = th + dt * ndth;
        S s, c;
        sincos_(nth, s, c);
        xn[0] = c; xn[1] = s; xn[2] = ndth;
    }
};


// deqmpc/rex_quadrotor.py:51-129 (RexQuadrotor_dynamics, default parameters): rigid body with
// position r (world), attitude as modified Rodrigues parameters m, body-frame velocity v and rate w;
// x = [r, m, v, w], u = four motor commands (scaled by act_scale = 100 inside the model), classic
// RK4 with the command held.  The module stores its constants as float32 tensors and promotes them
// when the state is double: the literals below are those float32 values.
struct RexQuadrotor {
    static constexpr int NX = 12, NU = 4;
    template <class S> __host__ __device__ static void cross3(const S *a, const S *b, S *o)
    {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    }
    // rotate r by the quaternion (qs, qv): rexquad_utils.py:211-220
    template <class S> __host__ __device__ static void quatrot(const S &qs, const S *qv, const S *r, S *o)
    {
        const S a = qs * qs - (qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2]);
        const S d = 2.0 * (qv[0] * r[0] + qv[1] * r[1] + qv[2] * r[2]);
        S c[3];
        cross3(qv, r, c);
        const S s2 = 2.0 * qs;
#pragma unroll
        for (int i = 0; i < 3; ++i) o[i] = a * r[i] + qv[i] * d + s2 * c[i];
    }
    // us = act_scale * u
    template <class S> __host__ __device__ static void deriv(const S *x, const S *us, S *dx)
    {
        constexpr double kf = 0.0244101, km = 0.00029958, bf = -30.48576, L = 0.28, mass = 2.0;
        constexpr double ssv = 0.7071067690849304;                       // float32(1/sqrt 2)
        constexpr double J00 = 1.5660889446735382e-02, J01 = 3.1803699584997958e-06, J11 = 1.5620780177414417e-02,
                         J22 = 2.2268680855631828e-02;
        constexpr double I00 = 6.3853336334228516e+01, I01 = -1.3000453822314739e-02, I11 = 6.4017295837402344e+01,
                         I22 = 4.4906116485595703e+01;
        constexpr double mg = -19.6200008392334, Bfz = -121.94303894042969;   // mass * float32(-9.81), float32(4 bf)
        const S *m = x + 3, *v = x + 6, *w = x + 9;
        // mrp -> quaternion (rexquad_utils.py:297-300)
        const S sq = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
        const S inv = S(1.0) / (1.0 + sq);
        const S qs = (1.0 - sq) * inv;
        S qv[3], qn[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { qv[i] = 2.0 * m[i] * inv; qn[i] = -qv[i]; }
        // forces (rex_quadrotor.py:51-68): thrust + gravity rotated into the body frame + motor bias
        S grav[3] = {S(0.0), S(0.0), S(mg)}, f[3];
        quatrot(qs, qn, grav, f);
        f[2] = f[2] + kf * (us[0] + us[1] + us[2] + us[3]) + Bfz;
        // moments (rex_quadrotor.py:70-85): arm x thrust of each motor + yaw drag torque
        S c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = kf * us[i] + bf;
        const double a = L * ssv;
        S tau[3];
        tau[0] = a * c[0] - a * c[1] - a * c[2] + a * c[3];              // sum_i (L ss_i)_y c_i
        tau[1] = -(a * c[0] + a * c[1] - a * c[2] - a * c[3]);           // -sum_i (L ss_i)_x c_i
        tau[2] = km * us[0] - km * us[1] + km * us[2] - km * us[3];
        // kinematics (rexquad_utils.py:393-402) and rigid-body equations (rex_quadrotor.py:114-129)
        quatrot(qs, qv, v, dx);
        const S m00 = m[0] * m[0], m11 = m[1] * m[1], m22 = m[2] * m[2];
        dx[3] = 0.25 * ((1.0 + m00 - m11 - m22) * w[0] + 2.0 * (m[0] * m[1] - m[2]) * w[1] + 2.0 * (m[0] * m[2] + m[1]) * w[2]);
        dx[4] = 0.25 * (2.0 * (m[1] * m[0] + m[2]) * w[0] + (1.0 - m00 + m11 - m22) * w[1] + 2.0 * (m[1] * m[2] - m[0]) * w[2]);
        dx[5] = 0.25 * (2.0 * (m[2] * m[0] - m[1]) * w[0] + 2.0 * (m[2] * m[1] + m[0]) * w[1] + (1.0 - m00 - m11 + m22) * w[2]);
        S wv[3], Jw[3], wJw[3];
        cross3(w, v, wv);
#pragma unroll
        for (int i = 0; i < 3; ++i) dx[6 + i] = f[i] * (1.0 / mass) - wv[i];
        Jw[0] = J00 * w[0] + J01 * w[1];
        Jw[1] = J01 * w[0] + J11 * w[1];
        Jw[2] = J22 * w[2];
        cross3(w, Jw, wJw);
        const S t0 = tau[0] - wJw[0], t1 = tau[1] - wJw[1], t2 = tau[2] - wJw[2];
        dx[9] = I00 * t0 + I01 * t1;
        dx[10] = I01 * t0 + I11 * t1;
        dx[11] = I22 * t2;
    }
    template <class S> __host__ __device__ static void step(const S *x, const S *u, double dt, S *xn)
    {
        // the stage derivatives are folded into the running sum as they come (same association as
        // k1 + 2 k2 + 2 k3 + k4 left to right): three 12-vectors live instead of six -- with K forward-mode
        // seeds riding along that is the difference between fitting the register file and not
        S us[NU], k[NX], y[NX], acc[NX];
#pragma unroll
        for (int i = 0; i < NU; ++i) us[i] = 100.0 * u[i];
        const double h2 = 0.5 * dt;
        deriv(x, us, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = k[i]; y[i] = x[i] + h2 * k[i]; }
        deriv(y, us, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = acc[i] + 2.0 * k[i]; y[i] = x[i] + h2 * k[i]; }
        deriv(y, us, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = acc[i] + 2.0 * k[i]; y[i] = x[i] + dt * k[i]; }
        deriv(y, us, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = x[i] + (dt / 6.0) * (acc[i] + k[i]);
    }
};

}  // namespace dyn
}  // namespace dqp
#endif
