"""numpy restatement of the augmented-Lagrangian Newton step of the reference (TEST
INFRASTRUCTURE ONLY -- see oracle/dqp_oracle.c for the rules; nothing in diff-qp-mpc_amd/
imports this).

  qpth/al_utils.py:62-102     merit_grad_hessian   -> merit_grad(), hessian()
  qpth/al_utils.py:162-318    constraint_res_jac2 & co -> constraint_jacobian()
  qpth/al_utils.py:414-418    cholesky_ex + cholesky_solve -> newton_update()
  qpth/al_utils.py:477-480    backward solve       -> chol_solve_neg()
  deqmpc/envs.py:5-47,62-76   PendulumDynamics (+ Jacobian, analytic here) -> pendulum_step()

Pinned by tests/test_oracle_golden.py::test_al_* against tests/golden/AL_*.npz (outputs of the
reference itself, tests/golden/make_golden_al.py).
"""
import numpy as np

DT, G, M_, L_ = 0.05, 10.0, 1.0, 1.0


def pendulum_step(x, u):
    """semi-implicit Euler (envs.py:23-31): returns x_next (N,2), dfdx (N,2,2), dfdu (N,2,1)."""
    th, thd = x[:, 0], x[:, 1]
    acc = (u[:, 0] + M_ * G * L_ * np.sin(th)) / (M_ * L_ ** 2)
    nthd = thd + acc * DT
    nth = th + nthd * DT
    c = M_ * G * L_ * np.cos(th) / (M_ * L_ ** 2)
    dfdx = np.zeros((x.shape[0], 2, 2))
    dfdx[:, 1, 0] = c * DT
    dfdx[:, 1, 1] = 1.0
    dfdx[:, 0, 0] = 1.0 + c * DT * DT
    dfdx[:, 0, 1] = DT
    dfdu = np.zeros((x.shape[0], 2, 1))
    dfdu[:, 1, 0] = DT / (M_ * L_ ** 2)
    dfdu[:, 0, 0] = DT * DT / (M_ * L_ ** 2)
    return np.stack((nth, nthd), 1), dfdx, dfdu


def constraint_jacobian(xu, x0, u_lower, u_upper, step=pendulum_step):
    B, T, nt = xu.shape
    n = x0.shape[1]
    m = nt - n
    x, u = xu[:, :, :n], xu[:, :, n:]
    xn, fx, fu = step(x[:, :-1].reshape(-1, n), u[:, :-1].reshape(-1, m))
    xn, fx, fu = xn.reshape(B, T - 1, n), fx.reshape(B, T - 1, n, n), fu.reshape(B, T - 1, n, m)
    eq = np.concatenate((x[:, 1:] - xn, x[:, :1] - x0[:, None]), 1).reshape(B, -1)
    iq = np.concatenate((u - u_upper, u_lower - u), 2).reshape(B, -1)
    iqc = np.maximum(iq, 0.0)
    neq, nineq, nz = T * n, 2 * T * m, T * nt
    J = np.zeros((B, neq + nineq, nz))
    for t in range(T - 1):
        r, c = t * n, t * nt
        J[:, r:r + n, c:c + n] = -fx[:, t]
        J[:, r:r + n, c + n:c + nt] = -fu[:, t]
        J[:, r:r + n, c + nt:c + nt + n] = np.eye(n)
    J[:, (T - 1) * n:T * n, :n] = np.eye(n)
    for t in range(T):
        r, c = neq + t * 2 * m, t * nt + n
        J[:, r:r + m, c:c + m] = np.eye(m)
        J[:, r + m:r + 2 * m, c:c + m] = -np.eye(m)
    Jc = J.copy()
    Jc[:, neq:] *= (iqc > 0)[..., None]
    return np.concatenate((eq, iq), 1), np.concatenate((eq, iqc), 1), J, Jc


def merit_grad(xu, Qd, q, lam, rho, res_clamp, J, Jc):
    B = xu.shape[0]
    return ((Qd * xu + q).reshape(B, -1) + np.einsum("bc,bcn->bn", lam, J)
            + rho * np.einsum("bc,bcn->bn", res_clamp, Jc))


def hessian(Jc, Qd, rho):
    B = Jc.shape[0]
    H = np.einsum("bci,bcj->bij", Jc, Jc) * rho.reshape(B, 1, 1)
    idx = np.arange(Jc.shape[2])
    H[:, idx, idx] += Qd.reshape(B, -1)
    return H


def newton_update(Jc, Qd, rho, grad):
    """-> (update, L, info) with torch.linalg.cholesky_ex semantics for info."""
    H = hessian(Jc, Qd, rho)
    B, nz = grad.shape
    L = np.zeros_like(H); upd = np.full((B, nz), np.nan); info = np.zeros(B, dtype=np.int32)
    for i in range(B):
        try:
            L[i] = np.linalg.cholesky(H[i])
        except np.linalg.LinAlgError:
            for k in range(1, nz + 1):
                if np.linalg.eigvalsh(H[i][:k, :k]).min() <= 0:
                    info[i] = k
                    break
            continue
        y = np.linalg.solve(L[i], -grad[i])
        upd[i] = np.linalg.solve(L[i].T, y)
    return upd, L, info


def chol_solve_neg(L, rhs):
    return np.stack([-np.linalg.solve(L[i].T, np.linalg.solve(L[i], rhs[i])) for i in range(L.shape[0])])
