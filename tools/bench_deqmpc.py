#!/usr/bin/env python3
"""BASELINE config 5 call pattern on one GPU: a DEQ-MPC imitation-learning step (deqmpc/train.py:150-175:
deq_iter rounds of DEQLayer -> Tracking_MPC (AL_mpc on the cartpole-2 device model) -> L1 loss over all
rounds, backward through the solvers, flat gradient all-reduce, Adam step) on this rank's share of the
batch (BATCH, default 8192 = 65536 / 8).  With torchrun (RANK/WORLD_SIZE in the env) the gradient
all-reduce runs over RCCL; stand-alone it is skipped.

    python tools/bench_deqmpc.py                       # one GPU
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_deqmpc.py
"""
import argparse, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from diff_qp_mpc_amd import policies
from diff_qp_mpc_amd.dynamics import DeviceDynamics

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
if world > 1:
    # DQP_BENCH_ONE_DEVICE=1 DQP_BENCH_BACKEND=gloo: rehearse the N > 1 code path with all ranks on GPU 0
    one = os.environ.get("DQP_BENCH_ONE_DEVICE") == "1"
    torch.cuda.set_device(0 if one else local)
    dist.init_process_group(os.environ.get("DQP_BENCH_BACKEND", "nccl"))
robot = os.environ.get("ROBOT", "cartpole2l")
B, T, deq_iter = int(os.environ.get("BATCH", 8192)), int(os.environ.get("T", 5)), int(os.environ.get("DEQ_ITER", 6))
dyn = DeviceDynamics(robot, dt=0.03 if robot == "cartpole2l" else 0.05)
nx, nu = dyn.n_state, dyn.n_ctrl
ub = 250.0 if robot == "cartpole2l" else 100.0
env = types.SimpleNamespace(nx=nx, nu=nu, nq=nx // 2, dt=dyn.dt, dynamics=dyn, dynamics_derivatives=dyn.jac,
                            action_space=types.SimpleNamespace(high=np.array([ub] * nu), low=np.array([-ub] * nu)))
args = argparse.Namespace(T=T, nq=nx // 2, hdim=256, layer_type="mlp", deq_out_type=1, policy_out_type=1, deq_iter=deq_iter,
                          solver_type="al", qp_iter=1, eps=1e-2, warm_start=True, bsz=B,
                          Q=torch.ones(nx), R=1e-2 * torch.ones(nu), dtype="double", device="cuda")
torch.manual_seed(0)
policy = policies.DEQMPCPolicy(args, env)
opt = torch.optim.Adam(policy.model.parameters(), lr=1e-4)
gen = torch.Generator(device="cuda").manual_seed(rank)
x = (torch.rand(B, nx, device="cuda", generator=gen) - 0.5)
gs = x[:, None, :] * torch.linspace(1, 0, T, device="cuda")[None, :, None]
ga = torch.zeros(B, T, nu, device="cuda")
mask = torch.ones(B, T, device="cuda")
group = dist.group.WORLD if world > 1 else None
for _ in range(2):
    loss, _, _ = policies.train_step(policy, opt, x, gs, ga, mask, group=group)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
reps = int(os.environ.get("REPS", 5))
t0 = time.perf_counter()
for _ in range(reps):
    loss, _, _ = policies.train_step(policy, opt, x, gs, ga, mask, group=group)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
if rank == 0:
    print("DEQ-MPC train step %s  B=%d/GPU x %d GPU(s)  T=%d deq_iter=%d: %.1f ms per step, %.0f k trajectories/s (whole job), loss %.4f"
          % (robot, B, world, T, deq_iter, dt * 1e3, B * world / dt / 1e3, float(loss)))
if world > 1:
    dist.destroy_process_group()
