"""numpy restatement of one AL_mpc.MPC call of the reference -- al_solve with its NewtonAL iterations, the
20-candidate merit line search, the multiplier / penalty update and the implicit backward (TEST INFRASTRUCTURE
ONLY, like everything under oracle/: tests and bench.py's cpu_baseline leg are the only importers).

  qpth/AL_mpc.py:254-321      al_solve (cold start; the warm start of :270-276 in warm_start())
  qpth/al_utils.py:17-34      warm_start_al
  qpth/al_utils.py:37-59      merit_function
  qpth/al_utils.py:361-482    NewtonAL.forward / backward
  qpth/al_utils.py:503-527    line_search_newton
  qpth/al_utils.py:340-352    compute_cost (diagonal cost)

The constraint Jacobian, merit gradient and Hessian come from oracle/al_oracle.py (pinned by AL_*.npz).
`step(x (N,n), u (N,m)) -> (x_next, df/dx (N,n,n), df/du (N,n,m))` is the dynamics: oracle/dyn_ref.py (the
reference's CasADi C, oracle/_ref) for the cartpoles, tests/host (host build of the model templates, itself pinned
against the reference's outputs by DYN_*.npz) for the quadrotor.

Pinned by tests/test_oracle_golden.py::test_al_solve_oracle_* against CFG3 / CFG4 / CFG5 goldens (outputs of the
reference's AL_mpc.MPC).
"""
import numpy as np

from . import al_oracle

N_LS = 20
NEWTON_STEPS = 4


def compute_cost(xu, Qd, q):
    return (0.5 * (xu * Qd * xu).sum(-1) + (q * xu).sum(-1)).sum(-1)


def residuals(xu, x0, u_lower, u_upper, step):
    """(res, res_clamp) of al_utils.dyn_res: [x_{t+1} - f(x_t, u_t) (t < T-1); x_0 - x0 | u - u_upper, u_lower - u]."""
    B, T, nt = xu.shape
    n = x0.shape[1]
    m = nt - n
    x, u = xu[:, :, :n], xu[:, :, n:]
    xn = step(np.ascontiguousarray(x[:, :-1]).reshape(-1, n), np.ascontiguousarray(u[:, :-1]).reshape(-1, m))[0]
    eq = np.concatenate((x[:, 1:] - xn.reshape(B, T - 1, n), x[:, :1] - x0[:, None]), 1).reshape(B, -1)
    iq = np.concatenate((u - u_upper, u_lower - u), 2).reshape(B, -1)
    return np.concatenate((eq, iq), 1), np.concatenate((eq, np.maximum(iq, 0.0)), 1)


def merit(xu, Qd, q, x0, lam, rho, u_lower, u_upper, step):
    res, resc = residuals(xu, x0, u_lower, u_upper, step)
    return compute_cost(xu, Qd, q) + 0.5 * rho[:, 0] * (resc * resc).sum(1) + (lam * res).sum(1)


def newton_al(xu, x0, lam, rho, Qd, q, u_lower, u_upper, step):
    """NewtonAL.forward: four Newton steps on the augmented Lagrangian, each followed by the 20-candidate line
    search.  -> (x_est, L of the last step -- or its Hessian H once a Cholesky factorisation failed --, status,
    chol_fail)."""
    B, T, nt = xu.shape
    n = x0.shape[1]
    x_est = xu.copy()
    mer = merit(x_est, Qd, q, x0, lam, rho, u_lower, u_upper, step)
    L = status = H = None
    chol_fail = False
    for _ in range(NEWTON_STEPS):
        res, resc, J, Jc = al_oracle.constraint_jacobian(x_est, x0, u_lower, u_upper, step)
        grad = ((Qd * x_est + q).reshape(B, -1) + np.matmul(lam[:, None, :], J)[:, 0]
                + rho * np.matmul(resc[:, None, :], Jc)[:, 0])
        H = np.matmul(Jc.transpose(0, 2, 1), Jc) * rho[:, :, None]
        idx = np.arange(T * nt)
        H[:, idx, idx] += Qd.reshape(B, -1)
        if not chol_fail:
            try:
                L = np.linalg.cholesky(H)
                y = np.linalg.solve(L, -grad[:, :, None])
                update = np.linalg.solve(L.transpose(0, 2, 1), y)[:, :, 0].reshape(B, T, nt)
            except np.linalg.LinAlgError:
                chol_fail = True        # al_utils.py:419-427: NaN update -> LU for the whole batch, this step and after
        if chol_fail:
            L = None
            update = np.linalg.solve(H, -grad[:, :, None])[:, :, 0].reshape(B, T, nt)
        # line_search_newton: candidates x + 2^-k update, the first knot's state pinned to x0
        best = np.full(B, np.inf)
        x_best = x_est.copy()
        for k in range(N_LS):
            cand = x_est + (2.0 ** -k) * update
            cand[:, 0, :n] = x0
            mk = merit(cand, Qd, q, x0, lam, rho, u_lower, u_upper, step)
            take = mk < best            # torch.min keeps the first minimum
            best = np.where(take, mk, best)
            x_best[take] = cand[take]
        status = best < mer
        x_est = np.where(status[:, None, None], x_best, x_est)
        mer = best                      # `merit = new_merit` whether or not the step was accepted
    return x_est, (H if chol_fail else L), status, chol_fail


def warm_start(lam, cost_start, cost_hist, lam_hist, rho_hist):
    """al_utils.warm_start_al: the stored AL iterate (newest first) whose cost was already below the new start."""
    B = lam.shape[0]
    idx = np.argmax(cost_hist < cost_start[None], axis=0)
    b = np.arange(B)
    lh = lam_hist[idx, b]
    lam = lam * (np.linalg.norm(lh, axis=-1) / np.linalg.norm(lam, axis=-1))[:, None]
    return lam, rho_hist[idx, b]


def al_solve(x, u, x0, Qd, q, u_lower, u_upper, step, lam, rho, al_iter=2, history=None):
    """AL_mpc.MPC.al_solve.  lam (B, T n + 2 T m), rho (B, 1).  `history` = (cost_hist, lam_hist, rho_hist) lists of
    the previous call (oldest first, as the reference stores them) for the warm start, None after reinitialize().
    -> dict(x, u, lam, rho, L, history)."""
    B, T, n = x.shape
    xu = np.concatenate((x, u), 2)
    cost_start = compute_cost(xu, Qd, q)
    if history is not None:
        ch, lh, rh = (np.stack(h[::-1], 0) for h in history)
        lam, rho = warm_start(lam, cost_start, ch, lh, rh)
    hist = [[cost_start], [lam], [rho]]
    neq = T * n
    L, chol_fail = None, False
    for _ in range(al_iter):
        xu, L, _, chol_fail = newton_al(xu, x0, lam, rho, Qd, q, u_lower, u_upper, step)
        res, _ = residuals(xu, x0, u_lower, u_upper, step)
        lam = lam + rho * res
        lam = np.concatenate((lam[:, :neq], np.maximum(lam[:, neq:], 0.0)), 1)
        rho = rho * 10.0
        hist[0].append(compute_cost(xu, Qd, q)); hist[1].append(lam); hist[2].append(rho)
    return dict(x=xu[:, :, :n], u=xu[:, :, n:], xu=xu, lam=lam, rho=rho, L=L, chol_fail=chol_fail, history=hist)


def backward(L, xu, grad_xu, chol_fail=False):
    """NewtonAL.backward: inp_grad = -H^-1 g through the last Cholesky factor (or, after a Cholesky failure, an LU
    solve with the last Hessian, passed as L); dQ = inp_grad * x, dq = inp_grad."""
    B = xu.shape[0]
    g = grad_xu.reshape(B, -1, 1)
    if chol_fail:
        inp = -np.linalg.solve(L, g)[:, :, 0].reshape(xu.shape)
    else:
        y = np.linalg.solve(L, g)
        inp = -np.linalg.solve(L.transpose(0, 2, 1), y)[:, :, 0].reshape(xu.shape)
    return inp * xu, inp
