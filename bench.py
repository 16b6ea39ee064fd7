#!/usr/bin/env python
"""Headline benchmark: images/s of the ResNet18 SSL_CR (teacher-student consistency) training step on synthetic
256x256 uint8 patches, bf16 engine mode, one process per GPU.

  python bench.py [--gpus N --steps K --warmup W]                      (N=1 by default)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1, RCCL over xGMI)

A step = one iteration of eval_BreastPathQ_SSL_CR.train() at the per-GPU shapes of BASELINE config 4
(`--batch_size 512 --mu 7` over 8 GPUs -> b=64/GPU: 192 labeled + 448 strong-unlabeled student images, 448 weak-unlabeled
teacher images = 1088 distinct patches), full fine-tune (--modules_student 0), MSE+MSE loss, Adam: teacher eval forward,
student train-mode forward, losses, full backward, gradient all-reduce (N>1), fused Adam update.  Weak scaling: the
per-GPU batch is fixed.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_FWD = 2 * 2368733184            # backbone forward FLOPs per 256x256 image
F_BWD_FULL = 2 * F_FWD - 0.308e9  # + dgrad + wgrad, no dgrad for conv1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ssl_cr", choices=["ssl_cr", "fwd", "rsp"])
    ap.add_argument("--batch_size", type=int, default=64, help="per-GPU --batch_size b (ssl_cr: 3b labeled + 7b unlabeled)")
    ap.add_argument("--mu", type=int, default=7)
    ap.add_argument("--image_size", type=int, default=256)
    ap.add_argument("--modules_student", type=int, default=0, help="0 = full fine-tune (headline); 60 = reference default freeze")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--bn-sync", type=int, default=1, choices=[0, 1],
                    help="N>1: 1 = train-mode BatchNorm on global-batch statistics (RCCL all-reduce of per-channel sums, the "
                         "north-star form); 0 = per-replica statistics like the reference's nn.DataParallel (gradient buckets only)")
    ap.add_argument("--aux-stream", type=int, default=0, choices=[0, 1],
                    help="1 = teacher forward on a second HIP stream next to the student forward (about -2 %% step time); off by "
                         "default because concurrent launches make the per-kernel durations of the roofline leg meaningless")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def synth_u8(shape, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(0, 256, shape, dtype=torch.uint8, generator=g).to(device)


def build_nets(args, device, classes=1, triplet=False):
    from ssl_cr_histo_amd import net
    torch.manual_seed(42)                                   # the reference's default --seed
    if triplet:
        model, cls = net.TripletNet("resnet18"), net.Classifier(768, 6)
    else:
        model, cls = net.TripletNet_Finetune("resnet18"), net.FinetuneResNet(classes)
    return model.to(device), cls.to(device)


def cpu_baseline(args):
    """oracle (CPU restatement of the reference step, torch CPU fp32) timed on this box's host cores, bounded sample."""
    from collections import OrderedDict
    from oracle import model as OM, steps as S
    # torch-CPU convolutions on a 17-image batch stop scaling (and collapse) far below a 256-thread host: use 32 threads
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    hw, b, mu = args.image_size, 1, args.mu
    nx, nu = 3 * b, mu * b
    sd = OM.init_state(42, OM.net_param_specs())
    csd = OM.init_state(43, OM.classifier_param_specs("finetune", 1))
    def mk():
        p, bufs = OM.split_state(OrderedDict((k, v.clone()) for k, v in sd.items()))
        pc, _ = OM.split_state(OrderedDict((k, v.clone()) for k, v in csd.items()))
        p.update(pc)
        return p, bufs
    ps, bs = mk()
    pt, bt = mk()
    for i, v in enumerate(ps.values()):
        v.requires_grad_(i >= args.modules_student)
    opt = S.Adam(ps.values(), 1e-4, (0.9, 0.999), 1e-8, 1e-4)
    g = torch.Generator().manual_seed(1234)
    x = torch.randint(0, 256, (nx, 3, hw, hw), generator=g).float()
    u_w = torch.randint(0, 256, (nu, 3, hw, hw), generator=g).float()
    u_s = torch.randint(0, 256, (nu, 3, hw, hw), generator=g).float()
    y = torch.rand(nx, generator=g)
    times = []
    budget_t0 = time.time()
    for it in range(3):
        t0 = time.time()
        S.ssl_cr_step("mse", ps, bs, pt, bt, opt, x, y, u_w, u_s, 1.0, faithful=True)
        times.append(time.time() - t0)
        if it >= 1 and time.time() - budget_t0 > 30.0:       # bounded: ~10-30 s of CPU work
            break
    t = min(times[1:]) if len(times) > 1 else times[0]
    patches = nx + 2 * nu
    return {"value": round(patches / t, 2), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"oracle ssl_cr_step (faithful: 3 backbone passes per image like models/net.py:88-90), b={b} mu={mu} "
                      f"-> {nx} labeled + {nu}+{nu} unlabeled {hw}x{hw} patches, modules_student={args.modules_student}, "
                      f"fp32 torch-CPU, {threads} threads of {os.cpu_count()} host cores, best of {max(1, len(times) - 1)} timed "
                      f"step(s) after 1 warm-up ({t:.2f} s/step)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    from ssl_cr_histo_amd import dist as sdist
    from ssl_cr_histo_amd import engine as E
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
    eng = E.set_engine(E.Engine(device, args.dtype))
    sdist.attach_engine(eng)
    eng.set_bn_sync(bool(args.bn_sync))
    eng.set_aux_stream(bool(args.aux_stream))

    hw, b, mu = args.image_size, args.batch_size, args.mu
    lr, wd = 1e-4, 1e-4
    if args.workload == "ssl_cr":
        nx, nu = 3 * b, mu * b
        mt, ct = build_nets(args, device)
        ms, cs = build_nets(args, device)
        for m in (mt, ct):
            m.eval()
        for m in (ms, cs):
            m.train()
        for p in list(mt.parameters()) + list(ct.parameters()):
            p.requires_grad = False
        for i, (_, p) in enumerate(ms.named_parameters()):
            p.requires_grad = i >= args.modules_student
        te, st = eng.bind(mt, ct), eng.bind(ms, cs)
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=lr,
                               betas=(0.9, 0.999), weight_decay=wd)
        x = synth_u8((nx, 3, hw, hw), 1234 + rank, device)
        u_w = synth_u8((nu, 3, hw, hw), 2234 + rank, device)
        u_s = synth_u8((nu, 3, hw, hw), 3234 + rank, device)
        y = torch.rand(nx, generator=torch.Generator().manual_seed(4234 + rank)).to(device)
        patches = nx + 2 * nu

        def step():
            r = eng.step_ssl_cr(te, st, "mse", x, y, u_w, u_s, 1.0)
            st.optimizer_step(opt)
            return r
        bwd = args.modules_student < 60
        flops_step = nu * F_FWD + (nx + nu) * (F_FWD + (F_BWD_FULL if args.modules_student == 0 else 0))
        cfg = {"workload": f"eval_BreastPathQ_SSL_CR.train step, per-GPU --batch_size {b} --mu {mu}: student {nx}+{nu}, teacher {nu} "
                           f"({patches} distinct {hw}x{hw} uint8 patches/step/GPU), modules_student={args.modules_student}, Adam",
               "global_batch_patches": patches * world, "parallelism": f"dp{world}", "backward": bwd,
               "bn_sync": bool(args.bn_sync) if world > 1 else None, "aux_stream": bool(args.aux_stream)}
    elif args.workload == "fwd":
        n = 4 * b
        ms, cs = build_nets(args, device)
        ms.eval()
        st = eng.bind(ms, cs)
        x = synth_u8((n, 3, hw, hw), 1234 + rank, device)
        patches = n

        def step():
            return st.forward((x,), train=False)
        flops_step = n * F_FWD
        cfg = {"workload": f"ResNet18 TripletNet_Finetune forward-only (eval BN folded), N={n} {hw}x{hw}", "parallelism": f"dp{world}"}
    else:
        B = 2 * b
        ms, cs = build_nets(args, device, triplet=True)
        ms.train()
        st = eng.bind(ms, cs)
        opt = torch.optim.SGD(list(ms.parameters()) + list(cs.parameters()), lr=0.01, momentum=0.9, weight_decay=wd, nesterov=True)
        xs = [synth_u8((B, 3, hw, hw), 1234 + 10 * i + rank, device) for i in range(3)]
        y = torch.randint(0, 6, (B,), generator=torch.Generator().manual_seed(5234 + rank)).to(device)
        patches = 3 * B

        def step():
            r = eng.step_supervised(st, "ce", xs, y, train=True)
            st.optimizer_step(opt)
            return r
        flops_step = 3 * B * (F_FWD + F_BWD_FULL)
        cfg = {"workload": f"pretrain_BreastPathQ.train step (RSP), B={B} triplets {hw}x{hw}, SGD-Nesterov", "parallelism": f"dp{world}"}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = patches * world * args.steps / dt

    out = {"metric": "images/sec (ResNet18 SSL_CR step, 256x256 bf16 synthetic patches; whole job)" if args.workload == "ssl_cr"
           else f"images/sec ({args.workload})",
           "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic", "config": cfg,
           "per_gpu_images_per_s": round(value / world, 1),
           "achieved_tflops_per_gpu_algorithmic": round(flops_step / (ms_per_step * 1e-3) / 1e12, 2)}

    if rank == 0 and not args.no_roofline:
        # roofline leg: same steps again with every conv launch bracketed by HIP events on its own stream
        eng.profile(True)
        for _ in range(max(2, min(5, args.steps))):
            step()
        torch.cuda.synchronize()
        rows = eng.profile_table()          # per kernel template instance, sorted by total time
        eng.profile(False)
        peak = 2500.0 if args.dtype == "bf16" else 157.3
        nprof = max(2, min(5, args.steps))
        hbm_rows = [r for r in rows if r["flops"] == 0]
        rows = [r for r in rows if r["flops"] > 0]

        def roof(d):
            common = {"kernel": d["name"], "launches": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
                      "algorithmic_mb_per_launch": round(d["bytes"] / d["launches"] / 1e6, 2), "traffic": None,
                      "time_share_of_step": round(d["ms"] / (ms_per_step * nprof), 3)}
            if d["flops"] > 0:
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                return dict(bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                            algorithmic_gflop_per_launch=round(d["flops"] / d["launches"] / 1e9, 3), **common)
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            return dict(bound="hbm", achieved=round(ach, 1), peak=8000.0, unit="GB/s", frac=round(ach / 8000.0, 4), **common)
        if hbm_rows:
            out["roofline_hbm"] = roof(hbm_rows[0])       # the largest HBM-bound kernel (BatchNorm backward apply)
        if rows:
            d = rows[0]                       # the dominant MFMA kernel of the step
            out["roofline"] = roof(d)
            if hbm_rows and hbm_rows[0]["ms"] > d["ms"]:
                out["roofline_note"] = ("by total time the HBM-bound bn_bwd_apply kernel edges out the largest conv kernel; both "
                                        "roofs are reported (roofline = MFMA kernel, roofline_hbm = that kernel)")
            # HBM traffic of this kernel comes from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), which
            # cannot run inside this process: report the committed measurement of one instance beside its algorithmic bytes
            pmc = os.path.join(ROOT, "profiles", "r01_g_pmc_traffic.json")
            if os.path.exists(pmc):
                for e in json.load(open(pmc))["kernels"]:
                    if e["match"] in d["name"]:
                        # ratio = PMC traffic / algorithmic bytes over the shapes this kernel runs in the step, weighted by
                        # the step's launch mix; the launches differ from the measured ones only in N (640 vs 448)
                        out["roofline"]["traffic"] = round(e["ratio"] * out["roofline"]["algorithmic_mb_per_launch"] * 1e6)
                        out["roofline"]["traffic_pmc"] = {
                            "ratio": e["ratio"], "shapes": e["shapes"], "launch_mix": e["bench_mix"],
                            "unit_of_traffic": "bytes per launch = ratio x algorithmic bytes of the average launch",
                            "source": "profiles/r01_g_pmc_traffic.json"}
                        break
            out["conv_kernels"] = [{"kernel": r["name"], "launches_per_step": r["launches"] // nprof,
                                    "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                                    "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1),
                                    "ms_per_step": round(r["ms"] / nprof, 3)} for r in rows]
            tot_ms = sum(r["ms"] for r in rows)
            tot_fl = sum(r["flops"] for r in rows)
            out["conv_all"] = {"tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 1), "frac_of_peak": round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4),
                               "time_share_of_step": round(tot_ms / (ms_per_step * nprof), 3)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "ssl_cr":
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
