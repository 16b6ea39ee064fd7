"""CPU, world_size 2, gloo: the data-parallel recipe the engine implements with RCCL -- shard the batch, scale each rank's
loss by the GLOBAL count, all-reduce SUM gradients, and all-reduce BatchNorm (sum, sumsq) -- reproduces the single-process
result.  The compute here is the test oracle's (plain torch CPU); the engine's kernels are covered by the -m gpu tests."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ssl_cr_histo_amd import dist as sd
    r, w, _ = sd.init_process_group("gloo")
    assert (r, w) == (rank, world)
    # 1. unique-id style broadcast
    payload = sd.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
    assert payload == bytes(range(128))
    # 2. sharded loss/gradient == global
    torch.manual_seed(0)
    X, Y = torch.randn(10, 16), torch.randn(10)
    W = torch.randn(16, requires_grad=True)
    lo, hi = sd.shard_range(10, rank, world)
    xs, ys = X[lo:hi], Y[lo:hi]
    local = ((xs @ W - ys) ** 2).sum() / 10.0                 # scaled by the GLOBAL count, like sslcr_loss_desc.inv_nx_global
    local.backward()
    g = W.grad.clone()
    dist.all_reduce(g)                                        # SUM, no division afterwards
    total = sd.global_mean_of_scaled(local.detach())
    Wr = W.detach().clone().requires_grad_(True)
    ref = F.mse_loss(X @ Wr, Y)
    ref.backward()
    assert torch.allclose(g, Wr.grad, atol=1e-6) and abs(float(total) - float(ref)) < 1e-6
    # 3. synced BatchNorm: all-reduced (sum, sumsq) give the global-batch statistics and the same normalised output
    A = torch.randn(8, 4, 5, 5) * 3 + 1
    a = sd.shard_batch(A, rank, world)
    sums = torch.stack([a.sum((0, 2, 3)), (a * a).sum((0, 2, 3))]).double()
    dist.all_reduce(sums)
    cnt = A.numel() / 4
    mean = sums[0] / cnt
    var = sums[1] / cnt - mean * mean
    y = (a - mean.float().view(1, -1, 1, 1)) / torch.sqrt(var.float().view(1, -1, 1, 1) + 1e-5)
    yref = sd.shard_batch(F.batch_norm(A, None, None, None, None, True, 0.1, 1e-5), rank, world)
    assert torch.allclose(y, yref, atol=1e-5)
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_two_rank_gloo_data_parallel_recipe():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]
