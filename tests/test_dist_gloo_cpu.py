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
    assert calls == [(6,), (2, 4)], calls                     # the row-count header (3 digits + squares), then ONE collective for both steps
    assert abs(out["loss"].avg - (0.9 + 1.8) / 2) < 1e-6 and abs(out["acc"].avg - (6.0 / 8 + 3.0 / 8) / 2) < 1e-6
    m.add(share[0], 4)
    out = m.meters()                                          # a later read reduces only the new row
    assert calls == [(6,), (2, 4), (6,), (1, 4)] and abs(out["loss"].avg - (0.9 + 1.8 + 0.9) / 3) < 1e-6
    # ranks that gathered different numbers of steps (loaders of different length): the header catches it on EVERY rank before
    # the rows -- buffers of different lengths -- would meet in a collective
    for _ in range(rank + 1):
        m.add(share[0], 4)
    try:
        m.meters()
        raise AssertionError("a row-count mismatch across ranks went unnoticed")
    except RuntimeError as e:
        assert "different numbers of steps" in str(e)
    assert calls[-1] == (6,)                                  # only the header was reduced
    # 5. the engine side of the sharded path with world = 2: dist.attach_engine -> Engine.init_comm (unique id made on rank 0,
    #    broadcast over the process group, sslcr_comm_init with this rank / world on every rank) -> steps._meters(engine) ->
    #    Engine.all_reduce_sum -> sslcr_comm_all_reduce_f32.  No GPU here, so the four C-ABI entry points are a stub with the
    #    header's signatures (include/sslcr.h) whose all-reduce runs over gloo; the engine object is the real class.
    import ctypes
    from ssl_cr_histo_amd import _lib as L
    from ssl_cr_histo_amd import engine as E
    from ssl_cr_histo_amd import steps as ST
    log = []

    class Stub:
        def sslcr_comm_unique_id(self, buf):
            ident = bytes((7 * i + 3) % 251 for i in range(256))      # two 128-byte ids, as sslcr_comm_unique_id documents
            ctypes.memmove(buf, ident, 256)
            log.append(("unique_id", rank))
            return 0

        def sslcr_comm_init(self, handle, buf, r, w):
            log.append(("init", bytes(buf), r, w))
            self.rank, self.world = r, w
            return 0

        def sslcr_comm_info(self, handle, pr, pw, pt):
            pr._obj.value, pw._obj.value, pt._obj.value = self.rank, self.world, 1
            return 0

        def sslcr_comm_all_reduce_f32(self, handle, ptr, n, stream):
            addr = ptr.value if isinstance(ptr, ctypes.c_void_p) else int(ptr)
            t = torch.frombuffer((ctypes.c_float * n).from_address(addr), dtype=torch.float32)
            dist.all_reduce(t)
            log.append(("all_reduce", n))
            return 0
    stub = Stub()
    L.lib = lambda: stub
    L.stream_ptr = lambda: None
    eng = E.Engine.__new__(E.Engine)                          # (Engine() itself refuses to exist without a GPU: tests/test_abi_cpu.py)
    eng.handle, eng.device, eng.rank, eng.world = ctypes.c_void_p(1), torch.device("cpu"), 0, 1
    assert sd.attach_engine(eng) is eng
    assert (eng.rank, eng.world) == (rank, world) and eng.comm_info() == (rank, world, "rccl")
    want_id = bytes((7 * i + 3) % 251 for i in range(256))
    assert [e for e in log if e[0] == "init"] == [("init", want_id, rank, world)]      # rank 1 received rank 0's id
    assert (("unique_id", 0) in log) == (rank == 0)           # only rank 0 asks the library for an id
    mm = ST._meters(eng, ["loss", "loss_x", "loss_u", "acc"])
    assert mm.world == world and mm.reduce is not None
    rows = sd.shard_batch(torch.tensor([[0.3, 0.1, 0.2, 2.0], [0.6, 0.2, 0.4, 1.0]]), rank, world)    # one step's share per rank
    mm.add(rows[0].clone(), 4)
    out = mm.meters()
    assert abs(out["loss"].avg - 0.9) < 1e-6 and abs(out["acc"].avg - 3.0 / 8) < 1e-6
    assert [e for e in log if e[0] == "all_reduce"] == [("all_reduce", 6), ("all_reduce", 4)]          # header + one row
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
