#!/usr/bin/env python3
"""Golden vectors for the DEQ-MPC call pattern (SURVEY.md §8 f4 / BASELINE config 5's loop):
deqmpc/policies.py DEQMPCPolicy.forward (:444-529) = deq_iter x [DEQLayer MLP -> Tracking_MPC
(AL_mpc.MPC)], the L1 imitation loss over every iterate (compute_loss_deqmpc, :800-808) and its
gradients wrt the DEQLayer parameters -- produced by importing the reference (build container
only) on its pendulum environment (deqmpc/envs.py PendulumEnv) with seeded weights.

Stored: the DEQLayer state_dict (so the mirror loads the very same weights), inputs (x, gt states /
actions / mask), every iterate's (network reference, MPC states, MPC actions), the loss and the
parameter gradients.

Usage:  python tests/golden/make_golden_deqmpc.py
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DQP_REFERENCE", "/root/reference")
m = types.ModuleType("ipdb")
def _st(*a, **k):
    raise RuntimeError("ipdb.set_trace() reached inside the reference")
m.set_trace = _st
sys.modules["ipdb"] = m
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "deqmpc"))

import envs  # noqa: E402
import policies  # noqa: E402

torch.manual_seed(0)
np.random.seed(0)
B, T, deq_iter, hdim = 6, 5, 3, 32
env = envs.PendulumEnv(stabilization=False)
env.nq = 1      # Tracking_MPC reads env.nq (policies.py:573); deqmpc/envs.py's pendulum (the one env that
                # imports without the compiled extensions) does not define it
args = argparse.Namespace(T=T, nq=1, hdim=hdim, layer_type="mlp", deq_out_type=1, policy_out_type=1,
                          deq_iter=deq_iter, solver_type="al", qp_iter=1, eps=1e-2, warm_start=True, bsz=B,
                          Q=env.Qlqr, R=env.Rlqr, dtype="double", device="cpu", kernel_width=3, pooling="mean",
                          deq=True, en_qp_solve=True)
policy = policies.DEQMPCPolicy(args, env)


def make_inputs(seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.stack([2.0 * (torch.rand(B, generator=g) - 0.5), torch.rand(B, generator=g) - 0.5], 1).float()
    gt_states = x[:, None, :].repeat(1, T, 1) * torch.linspace(1, 0, T)[None, :, None]
    gt_actions = 0.1 * torch.randn(B, T, 1, generator=g).float()
    mask = torch.ones(B, T).float()
    mask[0, -1] = 0.0
    return x, gt_states, gt_actions, mask


# The AL line search picks the argmin of 20 candidate steps; on a trajectory that sits in a flat
# direction of the merit the pick is decided by round-off and the reference's own output moves by
# ~1e-2 under a 1e-7 perturbation of x.  Such a batch pins nothing, so the input seed is advanced
# until the reference reproduces itself to 2e-5 under that perturbation.
for seed in range(1, 40):
    x, gt_states, gt_actions, mask = make_inputs(seed)
    with torch.no_grad():
        t1, _ = policy(x, gt_states, gt_actions, mask, qp_solve=True)
        t2, _ = policy(x + 1e-7, gt_states, gt_actions, mask, qp_solve=True)
    dev = max(float((a[1] - b[1]).abs().max()) for a, b in zip(t1, t2))
    print("input seed", seed, "self-deviation under a 1e-7 perturbation: %.2e" % dev)
    if dev < 2e-5:
        break
else:
    raise SystemExit("no stable batch found")
trajs, dyn_res = policy(x, gt_states, gt_actions, mask, qp_solve=True, lastqp_solve=False)
loss, loss_end = policies.compute_loss(policy, gt_states, gt_actions, mask, trajs, args)
policy.zero_grad()
loss.backward()
out = dict(x=x.numpy(), gt_states=gt_states.numpy(), gt_actions=gt_actions.numpy(), mask=mask.numpy(),
           loss=float(loss), loss_end=float(loss_end), Q=env.Qlqr.numpy(), R=env.Rlqr.numpy(),
           u_lower=env.action_space.low, u_upper=env.action_space.high, dt=env.dt,
           T=T, deq_iter=deq_iter, hdim=hdim, input_seed=seed)
for k, v in policy.model.state_dict().items():
    out["w_" + k] = v.numpy()
for k, p in policy.model.named_parameters():
    out["g_" + k] = p.grad.numpy() if p.grad is not None else np.zeros_like(p.detach().numpy())
for i, (net, xs, us) in enumerate(trajs):
    out["it%d_net" % i] = net.detach().numpy()
    out["it%d_x" % i] = xs.detach().numpy()
    out["it%d_u" % i] = us.detach().numpy()
np.savez_compressed(os.path.join(HERE, "DEQMPC_pendulum_T5_b6.npz"), **out)
print("loss", float(loss), "loss_end", float(loss_end), "dyn_res", dyn_res,
      "grad norm", float(sum((p.grad ** 2).sum() for p in policy.model.parameters() if p.grad is not None) ** 0.5))
