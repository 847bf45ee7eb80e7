"""Two ranks (torch.distributed, gloo, both on GPU 0 -- the box has one card) solve the two shards of an
MPC-structured batch through the product operators with sharding.global_batch_rule: the bitwise-OR
all_reduce of the batch rule's iteration masks makes every shard stop where the whole batch stops, so the
gathered solution equals the single-device solve of the whole batch bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import diff_qp_mpc_amd as dqp
    from diff_qp_mpc_amd import sharding
    from families import family_mpc
    ins = [torch.tensor(a, dtype=torch.float64, device="cuda") for a in family_mpc(3, B)]
    local, (lo, hi, nb) = sharding.shard_params(ins, (3, 2, 3, 2, 3, 2), world, rank)
    with sharding.global_batch_rule():
        z_local = dqp.QPFunction(verbose=-1, check_Q_spd=False)(*local)
    z_own = dqp.QPFunction(verbose=-1, check_Q_spd=False)(*local)            # shard-local rule, for contrast
    z = sharding.gather_solution(z_local.cpu(), nb)                            # gloo: gather on the host
    if rank == 0:
        whole = dqp.QPFunction(verbose=-1, check_Q_spd=False)(*ins)
        q.put((z.numpy(), whole.cpu().numpy(), float((z_own - whole[lo:hi]).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_global_batch_rule_bitwise():
    import socket
    with socket.socket() as sk:             # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    B, world = 90, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    z, whole, local_gap = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(z, whole)
    assert local_gap < 1e-6          # the shard-local rule agrees to the float tolerance only
