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
    total = local.detach().clone()
    dist.all_reduce(total)                                    # per-rank losses are scaled by 1/global-count: their SUM is the global mean
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
    # 4. the step functions' loss meters: per-rank shares stay local per step, ONE all-reduce when the meters are read
    #    (steps._Meters; on the GPU the reduce is the engine's sslcr_comm_all_reduce_f32, here gloo stands in)
    from ssl_cr_histo_amd.steps import _Meters
    calls = []

    def reduce(t):
        calls.append(tuple(t.shape))
        dist.all_reduce(t)
    m = _Meters(["loss", "loss_x", "loss_u", "acc"], reduce, world)
    share = [torch.tensor([0.3, 0.1, 0.2, 2.0]) * (rank + 1), torch.tensor([0.6, 0.2, 0.4, 1.0]) * (rank + 1)]
    for sh in share:
        m.add(sh, 4)                                          # 4 labeled images per rank and step -> global count 8
    out = m.meters()
    assert calls == [(2, 4)], calls                           # one collective for both steps
    assert abs(out["loss"].avg - (0.9 + 1.8) / 2) < 1e-6 and abs(out["acc"].avg - (6.0 / 8 + 3.0 / 8) / 2) < 1e-6
    m.add(share[0], 4)
    out = m.meters()                                          # a later read reduces only the new row
    assert calls == [(2, 4), (1, 4)] and abs(out["loss"].avg - (0.9 + 1.8 + 0.9) / 3) < 1e-6
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
