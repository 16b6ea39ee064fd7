#!/usr/bin/env python
"""Headline benchmark: images/s of the ResNet18 SSL_CR (teacher-student consistency) training step on synthetic
256x256 uint8 patches, bf16 engine mode, one process per GPU.

  python bench.py [--gpus N --steps K --warmup W]      N=1 by default; N>1 re-executes itself through
                                                       `python -m torch.distributed.run --nproc-per-node N` (one rank per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (what the driver runs for N>1)

A step = one iteration of eval_BreastPathQ_SSL_CR.train() at the per-GPU shapes of BASELINE config 4
(`--batch_size 512 --mu 7` over 8 GPUs -> b=64/GPU: 192 labeled + 448 strong-unlabeled student images, 448 weak-unlabeled
teacher images = 1088 distinct patches), full fine-tune (--modules_student 0), MSE+MSE loss, Adam: teacher eval forward,
student train-mode forward, losses, full backward, gradient all-reduce (N>1), fused Adam update.  Weak scaling: the
per-GPU batch is fixed.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0; at N=1 the line
also carries the other BASELINE.json configurations measured in the same process (`also`: forward-only = config 2, RSP = config 3,
the reference-default frozen backbone, and the fp32 exact-parity mode, config 5's shape in bf16 and fp8) and the CPU baseline of BASELINE.md section 3 (run alone, after the GPU
legs of record).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

_T0 = time.time()
import torch  # noqa: E402

# a fresh box pages the image in on first use (`import torch` then takes 1-2 minutes instead of 1.5 s) and the child processes of
# this run -- CPU baseline, rocprofv3 passes -- meet their own cold libraries: their wall bounds are stretched by what was seen here
_T_IMPORT = time.time() - _T0
COLD_EXTRA_S = min(300.0, 3.0 * _T_IMPORT) if _T_IMPORT > 15.0 else 0.0

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_FWD = 2 * 2368733184            # backbone forward FLOPs per 256x256 image
F_BWD_FULL = 2 * F_FWD - 0.308e9  # + dgrad + wgrad, no dgrad for conv1
PMC_FILE = os.path.join("profiles", "r05_final_pmc_step.json")     # fallback only (tools/pmc_step.sh on the builder's lease): the line's
                                                              # counters are measured IN this run by pmc_in_run() when rocprofv3 is there


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ssl_cr", choices=["ssl_cr", "fwd", "rsp", "cam_cr"],
                    help="ssl_cr = BreastPathQ consistency step (headline, config 4); cam_cr = Camelyon consistency step (config 5: CE + "
                         "hard-pseudo-label CE, two class loaders, SGD-Nesterov; use --batch_size 128 for its per-GPU shape)")
    ap.add_argument("--batch_size", type=int, default=64, help="per-GPU --batch_size b (ssl_cr: 3b labeled + 7b unlabeled)")
    ap.add_argument("--mu", type=int, default=7)
    ap.add_argument("--image_size", type=int, default=256)
    ap.add_argument("--modules_student", type=int, default=0, help="0 = full fine-tune (headline); 60 = reference default freeze")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="engine mode: bf16 (headline), fp32 (exact parity), fp8 = bf16 + e4m3 forward convs of layers 2-4 (config 5)")
    ap.add_argument("--bn-sync", type=int, default=1, choices=[0, 1],
                    help="N>1: 1 = train-mode BatchNorm on global-batch statistics (RCCL all-reduce of per-channel sums, the "
                         "north-star form); 0 = per-replica statistics like the reference's nn.DataParallel (gradient buckets only)")
    ap.add_argument("--aux-stream", type=int, default=-1, choices=[-1, 0, 1],
                    help="1 = teacher forward on a second HIP stream next to the student forward; -1 (default) = on when the job is "
                         "sharded (N > 1: the student's 34 BatchNorm all-reduces per step are latency-bound waits on its stream, which the "
                         "teacher's eval forward -- no collectives -- can fill), off at N = 1: round 6 measured both "
                         "side streams on five boxes (profiles/r06_streams_ab.txt) -- with them every box lands at 15.5-15.6 ms, without "
                         "them the same boxes run 15.2-15.75 ms: a gain on the slow boxes, a loss on the fast ones, zero on average.  The "
                         "roofline leg is always serialised (sslcr_profile forces both off)")
    ap.add_argument("--wgrad-stream", type=int, default=0, choices=[0, 1],
                    help="1 = backward's weight-gradient launches on a second HIP stream; bit-identical.  See --aux-stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the other configurations (forward-only, RSP, frozen, fp32 parity, config 5)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes (MFMA busy, HBM traffic per kernel)")
    ap.add_argument("--virtual-ranks", type=int, default=0,
                    help="W > 1: run the SHARDED job's control flow on ONE GPU -- W engine contexts (one host thread and one stream each) "
                         "that exchange through the engine's virtual communicator (sslcr_vcomm_*) instead of RCCL, through the same code "
                         "as --gpus W: workload, timed region, max over ranks, roofline leg on every rank, final barrier.  A check of "
                         "the multi-rank control flow (a rank-0-only step would hang here exactly as under RCCL), not a scaling number")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--also-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def synth_u8(shape, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(0, 256, shape, dtype=torch.uint8, generator=g).to(device)


def build_nets(device, classes=1, triplet=False):
    from ssl_cr_histo_amd import net
    torch.manual_seed(42)                                   # the reference's default --seed
    if triplet:
        model, cls = net.TripletNet("resnet18"), net.Classifier(768, 6)
    else:
        model, cls = net.TripletNet_Finetune("resnet18"), net.FinetuneResNet(classes)
    return model.to(device), cls.to(device)


# ------------------------------------------------------------------------------------------------ CPU baseline
CPU_BASELINE_WALL_S = 100.0 + COLD_EXTRA_S       # hard bound on the whole leg (the child process is killed at the deadline)


def cpu_baseline_child(args):
    """runs in a child process (`bench.py --cpu-baseline-child`): one JSON line per finished leg, most important legs first."""
    from collections import OrderedDict
    from oracle import model as OM, steps as S
    hw, b, mu = args.image_size, 2, args.mu
    nx, nu = 3 * b, mu * b
    patches = nx + 2 * nu
    sd = OM.init_state(42, OM.net_param_specs())
    csd = OM.init_state(43, OM.classifier_param_specs("finetune", 1))
    g = torch.Generator().manual_seed(1234)
    x = torch.randint(0, 256, (nx, 3, hw, hw), generator=g).float()
    u_w = torch.randint(0, 256, (nu, 3, hw, hw), generator=g).float()
    u_s = torch.randint(0, 256, (nu, 3, hw, hw), generator=g).float()
    y = torch.rand(nx, generator=g)

    def leg(threads, faithful, budget_s, warm=3, timed_steps=10):
        print(json.dumps({"started": True, "threads": threads, "variant": "faithful" if faithful else "algorithmic"}), flush=True)
        torch.set_num_threads(threads)

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
        times, t_begin = [], time.time()
        for it in range(warm + timed_steps):                           # 3 warm-up + >= 10 timed steps, median
            t0 = time.time()
            S.ssl_cr_step("mse", ps, bs, pt, bt, opt, x, y, u_w, u_s, 1.0, faithful=faithful)
            if it >= warm:
                times.append(time.time() - t0)
            if times and time.time() - t_begin > budget_s:            # bounded: stop adding steps once the share is used
                break
        times.sort()
        med = times[len(times) // 2]
        print(json.dumps({"threads": threads, "variant": "faithful" if faithful else "algorithmic",
                          "images_per_s": round(patches / med, 2), "s_per_step": round(med, 3), "timed_steps": len(times),
                          "warmup_steps": warm, "patches_per_step": patches}), flush=True)
    ncpu = os.cpu_count() or 1
    # torch-CPU convolutions on a 34-image batch do not scale with the thread count: measured on the MI355X box's 2 x EPYC 9575F
    # (256 logical cores) 0.29 s/step at 16 threads, 0.33 at 32, 0.69 at 64, 1.49 at 128, and not one step in 3 min at 256 -- so
    # the legs of record are 16 and 32 threads with BASELINE.md section 3's 3 warm-up + 10 timed steps (about 25 s together), then
    # one single-thread step for the per-core order of magnitude.  The half / all-core legs of round 2 are gone: they are slower
    # than 16 threads by 5x and more, and the all-core one never finished
    small = [t for t in (16, 32) if t <= ncpu] or [ncpu]
    for t in small:
        leg(t, True, 30.0)
        leg(t, False, 15.0)
    leg(1, False, 0.0, warm=0, timed_steps=1)     # one single-thread step, no warm-up: order of magnitude per core


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference step, torch-CPU fp32, proved equal to the reference's train() on the
    committed goldens) timed on this box's host cores: config C4 at b=2, mu=7 (student 20 / teacher 14 images, 34 distinct
    patches per step), full fine-tune, Adam -- FAITHFUL (three backbone passes per image, models/net.py:88-90: what the
    reference executes) and ALGORITHMIC (one pass), at 16 and 32 threads (3 warm-up steps, median of 10 timed
    steps) and one single-thread step of the algorithmic variant (more threads are slower on this batch, see cpu_baseline_child).
    Bounded: every leg stops adding steps once its share is used, and the whole thing runs in a child process that is killed at
    CPU_BASELINE_WALL_S (legs finished by then count).  `value` = the best faithful leg (the reference-equivalent baseline of
    record)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--image_size", str(args.image_size), "--mu", str(args.mu),
           "--modules_student", str(args.modules_student)]
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""                                   # the child never touches the GPU
    t0 = time.time()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
    timed_out = False
    try:
        stdout, _ = proc.communicate(timeout=CPU_BASELINE_WALL_S)
    except subprocess.TimeoutExpired:
        timed_out = True
        proc.kill()
        stdout, _ = proc.communicate()
    lines = [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]
    legs = [l for l in lines if not l.get("started")]
    if timed_out and lines and lines[-1].get("started"):              # the leg that was running when the wall bound hit
        legs.append({"threads": lines[-1]["threads"], "variant": lines[-1]["variant"], "finished": False,
                     "note": f"killed at the {CPU_BASELINE_WALL_S:.0f} s wall bound of the whole CPU leg"})
    ncpu = os.cpu_count() or 1
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    faithful = [l for l in legs if l["variant"] == "faithful" and l.get("finished", True)]
    if not faithful:
        return {"value": None, "unit": "images/s", "cores": 0, "kind": "port", "sample": "no leg finished inside the wall bound",
                "host": {"cpu": cpu_model, "logical_cores": ncpu}, "legs": legs, "timed_out": timed_out}
    best = max(faithful, key=lambda l: l["images_per_s"])
    b, mu, hw = 2, args.mu, args.image_size
    return {"value": best["images_per_s"], "unit": "images/s", "cores": best["threads"], "kind": "port",
            "sample": f"oracle ssl_cr_step (CPU restatement of eval_BreastPathQ_SSL_CR.train, torch-CPU fp32), config C4 at b={b} mu={mu}: "
                      f"{3 * b} labeled + {mu * b}+{mu * b} unlabeled {hw}x{hw} patches = {best['patches_per_step']} distinct patches/step, "
                      f"modules_student={args.modules_student}, Adam; value = FAITHFUL variant (3 backbone passes per image like "
                      f"models/net.py:88-90) at {best['threads']} threads, median of {best['timed_steps']} timed step(s) after "
                      f"{best['warmup_steps']} warm-up "
                      f"({best['s_per_step']} s/step); all legs in `legs`",
            "host": {"cpu": cpu_model, "logical_cores": ncpu}, "legs": legs, "wall_s": round(time.time() - t0, 1),
            "timed_out": timed_out}


# ------------------------------------------------------------------------------------------------ clock / power next to the roofline
class PowerSampler:
    """sclk and socket power of THIS process's GPU while a leg runs (a thread reading the amdgpu hwmon files every 20 ms; `rocm-smi
    --json` where they are not readable): the power-cap argument of DESIGN.md as numbers in the bench line.  Best effort: any failure
    leaves a note instead of numbers."""

    def __init__(self, device_index=0, period=0.02):
        import threading
        self.period, self.rows, self.note, self.hw = period, [], None, None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            import glob
            for card in glob.glob("/sys/class/drm/card*/device"):
                if bdf in os.path.realpath(card):
                    hm = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
                    if hm and os.path.exists(os.path.join(hm[0], "freq1_input")):
                        self.hw = hm[0]
            if self.hw is None:
                self.note = f"no hwmon directory for PCI device {bdf}*; rocm-smi --json instead"
        except Exception as e:                                  # noqa: BLE001
            self.note = f"hwmon lookup failed ({type(e).__name__}: {e}); rocm-smi --json instead"

    def _read(self):
        if self.hw:
            f = int(open(os.path.join(self.hw, "freq1_input")).read()) / 1e6
            pw = None
            for name in ("power1_input", "power1_average"):
                q = os.path.join(self.hw, name)
                if os.path.exists(q):
                    pw = int(open(q).read()) / 1e6
                    break
            return f, pw
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=5)
        d = next(iter(json.loads(r.stdout.strip().splitlines()[-1]).values()))
        f = float(d["sclk clock speed:"].strip("()Mhz"))
        pw = next((float(v) for k, v in d.items() if "Power (W)" in k), None)
        return f, pw

    def _run(self):
        while not self._stop.is_set():
            try:
                self.rows.append(self._read())
            except Exception as e:                              # noqa: BLE001
                self.note = f"sampling failed ({type(e).__name__}: {e})"
                return
            self._stop.wait(self.period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=10)
        return False

    def summary(self):
        fs = [r[0] for r in self.rows if r[0] is not None]
        ps = [r[1] for r in self.rows if r[1] is not None]
        out = {"samples": len(self.rows), "source": "amdgpu hwmon (freq1_input = sclk, power1_input)" if self.hw else "rocm-smi --json"}
        if fs:
            out.update(sclk_mhz_mean=round(sum(fs) / len(fs), 1), sclk_mhz_min=round(min(fs), 1), sclk_mhz_max=round(max(fs), 1))
        if ps:
            out.update(socket_power_w_mean=round(sum(ps) / len(ps), 1), socket_power_w_max=round(max(ps), 1))
        if self.hw and os.path.exists(os.path.join(self.hw, "power1_cap")):
            out["socket_power_cap_w"] = round(int(open(os.path.join(self.hw, "power1_cap")).read()) / 1e6, 1)
        if self.note:
            out["note"] = self.note
        return out


# ------------------------------------------------------------------------------------------------ workloads
def make_workload(name, eng, args, device, rank, world, modules_student=None):
    """-> (step(), distinct patches per step, algorithmic FLOPs per step, config dict)"""
    hw, b, mu = args.image_size, args.batch_size, args.mu
    ms_freeze = args.modules_student if modules_student is None else modules_student
    lr, wd = 1e-4, 1e-4
    if name == "ssl_cr":
        nx, nu = 3 * b, mu * b
        mt, ct = build_nets(device)
        ms, cs = build_nets(device)
        for m in (mt, ct):
            m.eval()
        for m in (ms, cs):
            m.train()
        for p in list(mt.parameters()) + list(ct.parameters()):
            p.requires_grad = False
        for i, (_, p) in enumerate(ms.named_parameters()):
            p.requires_grad = i >= ms_freeze
        te, st = eng.bind(mt, ct), eng.bind(ms, cs)
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=lr,
                               betas=(0.9, 0.999), weight_decay=wd)
        x = synth_u8((nx, 3, hw, hw), 1234 + rank, device)
        u_w = synth_u8((nu, 3, hw, hw), 2234 + rank, device)
        u_s = synth_u8((nu, 3, hw, hw), 3234 + rank, device)
        y = torch.rand(nx, generator=torch.Generator().manual_seed(4234 + rank)).to(device)

        def step():
            r = eng.step_ssl_cr(te, st, "mse", x, y, u_w, u_s, 1.0)
            st.optimizer_step(opt)
            return r
        flops = nu * F_FWD + (nx + nu) * (F_FWD + (F_BWD_FULL if ms_freeze == 0 else 0))
        cfg = {"workload": f"eval_BreastPathQ_SSL_CR.train step, per-GPU --batch_size {b} --mu {mu}: student {nx}+{nu}, teacher {nu} "
                           f"({nx + 2 * nu} distinct {hw}x{hw} uint8 patches/step/GPU), modules_student={ms_freeze}, Adam",
               "global_batch_patches": (nx + 2 * nu) * world, "parallelism": f"dp{world}", "backward": ms_freeze < 60,
               "bn_sync": bool(args.bn_sync) if world > 1 else None, "aux_stream": bool(args.aux_stream),
               "wgrad_stream": bool(args.wgrad_stream)}
        return step, nx + 2 * nu, flops, cfg, (mt, ct, ms, cs, opt)
    if name == "cam_cr":
        # eval_Camelyon_SSL_CR.train (:94-121): tumor + normal labeled loaders (b x 3 images each), tumor + normal unlabeled loaders (b x mu each)
        nx, nu = 2 * 3 * b, 2 * mu * b
        mt, ct = build_nets(device, classes=2)
        ms, cs = build_nets(device, classes=2)
        for m in (mt, ct):
            m.eval()
        for m in (ms, cs):
            m.train()
        for p in list(mt.parameters()) + list(ct.parameters()):
            p.requires_grad = False
        for i, (_, p) in enumerate(ms.named_parameters()):
            p.requires_grad = i >= ms_freeze
        te, st = eng.bind(mt, ct), eng.bind(ms, cs)
        opt = torch.optim.SGD(filter(lambda p: p.requires_grad, list(ms.parameters()) + list(cs.parameters())), lr=5e-4, momentum=0.9,
                              weight_decay=wd, nesterov=True)
        x = synth_u8((nx, 3, hw, hw), 1234 + rank, device)
        u_w = synth_u8((nu, 3, hw, hw), 2234 + rank, device)
        u_s = synth_u8((nu, 3, hw, hw), 3234 + rank, device)
        y = torch.randint(0, 2, (nx,), generator=torch.Generator().manual_seed(4234 + rank)).to(device)

        def step():
            r = eng.step_ssl_cr(te, st, "ce", x, y, u_w, u_s, 1.0)
            st.optimizer_step(opt)
            return r
        flops = nu * F_FWD + (nx + nu) * (F_FWD + (F_BWD_FULL if ms_freeze == 0 else 0))
        cfg = {"workload": f"eval_Camelyon_SSL_CR.train step, per-GPU --batch_size {b} --mu {mu}: student {nx}+{nu}, teacher {nu} "
                           f"({nx + 2 * nu} distinct {hw}x{hw} uint8 patches/step/GPU), modules_student={ms_freeze}, SGD-Nesterov",
               "global_batch_patches": (nx + 2 * nu) * world, "parallelism": f"dp{world}", "backward": ms_freeze < 60}
        return step, nx + 2 * nu, flops, cfg, (mt, ct, ms, cs, opt)
    if name == "fwd":
        n = 4 * b
        ms, cs = build_nets(device)
        ms.eval()
        st = eng.bind(ms, cs)
        x = synth_u8((n, 3, hw, hw), 1234 + rank, device)

        def step():
            return st.forward((x,), train=False)
        cfg = {"workload": f"ResNet18 TripletNet_Finetune forward-only (eval BN folded), N={n} {hw}x{hw}", "parallelism": f"dp{world}"}
        return step, n, n * F_FWD, cfg, (ms, cs)
    B = 2 * b
    ms, cs = build_nets(device, triplet=True)
    ms.train()
    st = eng.bind(ms, cs)
    opt = torch.optim.SGD(list(ms.parameters()) + list(cs.parameters()), lr=0.01, momentum=0.9, weight_decay=wd, nesterov=True)
    xs = [synth_u8((B, 3, hw, hw), 1234 + 10 * i + rank, device) for i in range(3)]
    y = torch.randint(0, 6, (B,), generator=torch.Generator().manual_seed(5234 + rank)).to(device)

    def step():
        r = eng.step_supervised(st, "ce", xs, y, train=True)
        st.optimizer_step(opt)
        return r
    cfg = {"workload": f"pretrain_BreastPathQ.train step (RSP), B={B} triplets {hw}x{hw}, SGD-Nesterov", "parallelism": f"dp{world}"}
    return step, 3 * B, 3 * B * (F_FWD + F_BWD_FULL), cfg, (ms, cs, opt)


def timed(step, warmup, steps, barrier, per_step=None):
    """W untimed steps, then exactly K steps between two barrier + synchronize points (host clock: the contract's number).
    per_step (a list): also gets each step's duration in ms from HIP events recorded on the launch stream between the steps."""
    for _ in range(warmup):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step is not None else None
    barrier()
    t0 = time.perf_counter()
    if ev:
        ev[0].record()
    for i in range(steps):
        step()
        if ev:
            ev[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    if ev:
        per_step.extend(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return dt


def also_child(args):
    """`bench.py --also-child`: forward-only (config 2), RSP (config 3), frozen backbone and the fp32 exact-parity mode; one JSON
    line."""
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    from ssl_cr_histo_amd import engine as E
    eng = E.set_engine(E.Engine(device, "bf16"))
    if args.aux_stream < 0:
        args.aux_stream = 0
    eng.set_aux_stream(bool(args.aux_stream))          # the side streams as in the headline run
    eng.set_wgrad_stream(bool(args.wgrad_stream))
    also = {"side_streams": {"aux_stream": bool(args.aux_stream), "wgrad_stream": bool(args.wgrad_stream)}}

    def barrier():
        torch.cuda.synchronize()

    def rec(tag, wl, eng_, steps, warmup, **kw):
        s, p, fl, c, k = make_workload(wl, eng_, args, device, 0, 1, **kw)
        t = timed(s, warmup, steps, barrier)
        peak = 2500.0 if eng_ is eng else 157.3
        also[tag] = {"images_per_s": round(p * steps / t, 1), "ms_per_step": round(t / steps * 1e3, 3), "steps": steps,
                     "achieved_tflops_algorithmic": round(fl / (t / steps) / 1e12, 2),
                     "frac_of_peak": round(fl / (t / steps) / 1e12 / peak, 4), "peak_tflops": peak, "workload": c["workload"]}
        del s, k
    rec("forward_only_config2", "fwd", eng, 20, 5)
    rec("rsp_config3", "rsp", eng, 20, 5)
    rec("frozen_backbone_modules_student_60", "ssl_cr", eng, 20, 5, modules_student=60)
    eng32 = E.Engine(device, "fp32")
    eng32.set_aux_stream(bool(args.aux_stream))
    eng32.set_wgrad_stream(bool(args.wgrad_stream))
    rec("parity_mode_fp32", "ssl_cr", eng32, 4, 2)
    also["parity_mode_fp32"]["note"] = ("exact-parity engine mode (v_mfma_f32_16x16x4_f32, fp32 storage): the mode that holds the "
                                        "north-star 1e-3 against the reference goldens; peak = 157.3 TF fp32 matrix")
    del eng32
    torch.cuda.empty_cache()
    # BASELINE config 5 at its per-GPU shape (eval_Camelyon_SSL_CR.py --batch_size 1024 over 8 GPUs -> b = 128 per class loader:
    # student 768 + 1792 = 2560, teacher 1792 images), in bf16 mode and in the fp8 mode (e4m3 forward convs of layers 2-4)
    import copy
    a5 = copy.copy(args)
    a5.batch_size = 128

    def rec5(tag, dtype):
        e5 = eng if dtype == "bf16" else E.Engine(device, dtype)
        e5.set_aux_stream(bool(args.aux_stream))
        e5.set_wgrad_stream(bool(args.wgrad_stream))
        s, p, fl, c, k = make_workload("cam_cr", e5, a5, device, 0, 1)
        n5 = 8
        t = timed(s, 3, n5, barrier)
        r = {"images_per_s": round(p * n5 / t, 1), "ms_per_step": round(t / n5 * 1e3, 3), "steps": n5, "dtype": dtype,
             "achieved_tflops_algorithmic": round(fl / (t / n5) / 1e12, 2), "workload": c["workload"]}
        e5.profile(True)
        for _ in range(2):
            s()
        torch.cuda.synchronize()
        rows = [q for q in e5.profile_table() if q["flops"] > 0]
        e5.profile(False)
        d = next((q for q in rows if "fp8" in q["name"]), rows[0]) if dtype == "fp8" else rows[0]
        pk = 5000.0 if "fp8" in d["name"] else 2500.0
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        r["roofline"] = {"bound": "mfma", "kernel": d["name"], "achieved": round(ach, 2), "peak": pk, "unit": "TFLOP/s",
                         "frac": round(ach / pk, 4), "launches": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
                         "time_share_of_step": round(d["ms"] / (r["ms_per_step"] * 2), 3)}
        also[tag] = r
        del s, k
        torch.cuda.empty_cache()
    rec5("bf16_config5", "bf16")
    rec5("fp8_config5", "fp8")
    print(json.dumps(also), flush=True)


def also_records(args):
    """The `also` legs run in a child process with the profiler's environment removed: under `rocprofv3 --kernel-trace --stats --
    python bench.py` the summary then covers exactly the headline workload (the legs launch the SAME kernels at other batch sizes;
    in one process their launches would be averaged into the headline's per-kernel durations)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--also-child", "--batch_size", str(args.batch_size), "--mu", str(args.mu),
           "--image_size", str(args.image_size), "--aux-stream", str(args.aux_stream), "--wgrad-stream", str(args.wgrad_stream)]
    env = _clean_profiler_env()
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    except subprocess.TimeoutExpired:
        return {"error": "the also legs did not finish in 600 s"}
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    if proc.returncode != 0 or not lines:
        return {"error": f"also child exited with {proc.returncode}", "stderr_tail": proc.stderr[-400:]}
    return json.loads(lines[-1])


def _clean_profiler_env():
    """this process's environment without a surrounding rocprofv3's hooks (children get their own profiler or none)"""
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith(("ROCPROF", "ROCP_", "ROCTX", "ROCTRACER")) or k in ("HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE"))}
    if "LD_PRELOAD" in env:
        kept = [x for x in env["LD_PRELOAD"].replace(":", " ").split() if "rocprof" not in x and "roctracer" not in x and "roctx" not in x]
        if kept:
            env["LD_PRELOAD"] = ":".join(kept)
        else:
            del env["LD_PRELOAD"]
    return env


PMC_PASSES = ["SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"]


def pmc_in_run(args):
    """MFMA-busy and HBM-traffic counters of THIS command's step, measured now: three counter-only `rocprofv3 --pmc ... --kernel-trace`
    passes (SQ + GRBM; FETCH_SIZE; WRITE_SIZE -- the TCC slots do not fit one pass, MI355X_MICROARCH.md) over a short child run of the
    same workload, reduced per kernel template instance by tools/pmc_step_reduce.py.  -> (document | None, note)"""
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_step_reduce
    out = tempfile.mkdtemp(prefix="sslcr_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "2", "--no-roofline", "--no-cpu-baseline", "--no-also",
             "--no-pmc", "--workload", args.workload, "--batch_size", str(args.batch_size), "--mu", str(args.mu), "--image_size",
             str(args.image_size), "--modules_student", str(args.modules_student), "--dtype", args.dtype,
             "--aux-stream", "0", "--wgrad-stream", "0"]        # per-kernel counters: one kernel at a time
    env = _clean_profiler_env()
    env["TMPDIR"] = "/tmp"
    t0 = time.time()
    try:
        for i, counters in enumerate(PMC_PASSES):
            cmd = [exe, "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", os.path.join(out, f"pass{i}"), "-o", "pmc",
                                                        "--"] + child
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd="/tmp", timeout=150 + COLD_EXTRA_S)
            if r.returncode != 0:
                return None, f"rocprofv3 pass {i} exited with {r.returncode}: {r.stdout[-300:]}"
        doc = pmc_step_reduce.reduce_dir(out, command="python bench.py " + " ".join(child[2:]))
    except subprocess.TimeoutExpired:
        return None, f"a rocprofv3 --pmc pass did not finish in {150 + COLD_EXTRA_S:.0f} s"
    finally:
        shutil.rmtree(out, ignore_errors=True)
    if not doc["kernels"]:
        return None, "the rocprofv3 passes produced no counter rows"
    doc["wall_s"] = round(time.time() - t0, 1)
    return doc, "ok"


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-execute through torch.distributed.run, one rank per GPU."""
    if torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


class _RealRanks:
    """one process per GPU (the contract's launch): torch.distributed for the barrier and the max over ranks"""

    def __init__(self, dist, device, world):
        self.dist, self.device, self.world = dist, device, world

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, rank, v):
        if self.world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


class _VirtualRanks:
    """W host threads of this process (--virtual-ranks W): a thread barrier stands in for dist.barrier"""

    def __init__(self, world):
        import threading
        self.world = world
        self._bar = threading.Barrier(world, timeout=300)
        self._vals = [0.0] * world

    def barrier(self):
        torch.cuda.current_stream().synchronize()
        self._bar.wait()
        torch.cuda.synchronize()

    def max_over_ranks(self, rank, v):
        self._vals[rank] = v
        self._bar.wait()
        m = max(self._vals)
        self._bar.wait()
        return m


def main():
    args = parse()
    if args.cpu_baseline_child:
        return cpu_baseline_child(args)
    if args.also_child:
        return also_child(args)
    if args.virtual_ranks > 1:
        return main_virtual(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world}")
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
    run_rank(args, rank, world, device, eng, _RealRanks(dist, device, world), "rccl")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_virtual(args):
    """--virtual-ranks W: the body of a rank (run_rank) on W threads of this process, one engine context and one stream each"""
    import threading
    from ssl_cr_histo_amd import engine as E
    world = args.virtual_ranks
    if args.gpus != 1:
        raise SystemExit("bench.py --virtual-ranks runs on one GPU (--gpus 1)")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    vc = E.VirtualComm(world)
    engines = [E.Engine(device, args.dtype) for _ in range(world)]
    for r, e in enumerate(engines):
        e.init_comm_virtual(vc, r, world)
    ranks = _VirtualRanks(world)
    streams = [torch.cuda.Stream(device=device) for _ in range(world)]
    err = [None] * world

    def body(r):
        try:
            torch.cuda.set_device(device)
            with torch.cuda.stream(streams[r]):
                run_rank(args, r, world, device, engines[r], ranks, "virtual")
                streams[r].synchronize()
        except BaseException as e:      # noqa: BLE001 -- reported below
            err[r] = e
            try:
                ranks._bar.abort()
            except Exception:           # noqa: BLE001
                pass
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(900)
    for e in err:
        if e is not None:
            raise e


def parity_record(out):
    """What the headline mode is held to (tests/measured_errors.json: errors measured on the MI355X against the REFERENCE's own
    full-size iteration, tests/golden/bpq_cr_full.npz) next to the mode that holds north_star's 1e-3."""
    rec = {"mode": out["dtype"], "north_star_1e-3_mode": "fp32",
           "note": "value is measured in `mode`.  Since round 6 (eval-mode BatchNorm scale kept out of the bf16 filters) the bf16 mode's LOSS "
                   "of this workload's full-size iteration is within 1e-3 of the reference (loss_rel_err_vs_reference); element-wise "
                   "logits / features are bf16 storage (1e-2) -- the 1e-3 bound on logits, features and trajectories holds in the fp32 "
                   "engine mode (--dtype fp32), whose throughput is fp32_images_per_s"}
    try:
        m = json.load(open(os.path.join(ROOT, "tests", "measured_errors.json")))
        k = "bf16" if out["dtype"] in ("bf16", "fp8") else out["dtype"]
        rec["loss_rel_err_vs_reference"] = m.get(f"bpq_cr_full/ret0/{k}")
        rec["feature_row_norm_rel_err_vs_reference"] = m.get(f"bpq_cr_full/feats_rowl2/{k}")
        rec["source"] = "tests/measured_errors.json (bpq_cr_full: this workload's iteration against the reference's, worst value recorded)"
    except OSError:
        rec["loss_rel_err_vs_reference"] = None
    fp32 = (out.get("also") or {}).get("parity_mode_fp32") if isinstance(out.get("also"), dict) else None
    rec["fp32_images_per_s"] = fp32.get("images_per_s") if isinstance(fp32, dict) else None
    return rec


def run_rank(args, rank, world, device, eng, ranks, want_transport):
    """one rank of the job, from the workload to the final barrier.  `world` > 1: every step is full of collectives (BatchNorm sums,
    gradient buckets), so EVERY rank runs every step of every leg; what only rank 0 does (rank0_only_legs) calls nothing collective."""
    eng.set_bn_sync(bool(args.bn_sync))
    if args.aux_stream < 0:
        args.aux_stream = 1 if (world > 1 and args.bn_sync) else 0
    eng.set_aux_stream(bool(args.aux_stream))
    eng.set_wgrad_stream(bool(args.wgrad_stream))
    barrier = ranks.barrier

    step, patches, flops_step, cfg, keep = make_workload(args.workload, eng, args, device, rank, world)
    per_step_ms = []
    # sclk and socket power of this GPU over the warm-up and the timed steps themselves (rank 0; a thread reading two sysfs files every
    # 20 ms): the clock the value was measured at.  (A separate one-second leg in front of the roofline leg was tried first: the part
    # settles ~5 % lower after a second of sustained load, which moved the per-kernel numbers of that leg away from the timed region's.)
    sampler = PowerSampler(device.index or 0) if rank == 0 else None
    if sampler:
        sampler.__enter__()
    dt = timed(step, args.warmup, args.steps, barrier, per_step_ms)
    power = None
    if sampler:
        sampler.__exit__()
        power = sampler.summary()
    dt = ranks.max_over_ranks(rank, dt)
    ms_per_step = dt / args.steps * 1e3
    value = patches * world * args.steps / dt
    crank, cworld, transport = eng.comm_info()
    if world > 1 and (cworld != world or transport != want_transport):
        raise SystemExit(f"bench.py: {world} ranks asked for, the engine's communicator reports {cworld} rank(s) over '{transport}' -- the "
                         f"{want_transport} communicator was not built; refusing to print a multi-rank number for independent replicas")

    out = {"metric": "images/sec (ResNet18 SSL_CR step, 256x256 bf16 synthetic patches; whole job)" if args.workload == "ssl_cr"
           else f"images/sec ({args.workload})",
           "value": round(value, 1), "unit": "images/s", "n_gpus": 1 if want_transport == "virtual" else world, "steps": args.steps,
           "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic", "config": cfg,
           "per_gpu_images_per_s": round(value / (1 if want_transport == "virtual" else world), 1),
           "achieved_tflops_per_gpu_algorithmic": round(flops_step * (world if want_transport == "virtual" else 1) / (ms_per_step * 1e-3) / 1e12, 2),
           "ranks_seen": cworld, "collective_transport": transport}
    if power:
        out["power_clock"] = power
    if want_transport == "virtual":
        out["virtual_ranks"] = world
        out["note"] = (f"{world} virtual ranks on ONE GPU (sslcr_vcomm): the sharded job's control flow, not a scaling number -- value is "
                       "the whole job's images/s on this one device")
    # each timed step by HIP events on the launch stream (this rank): median and spread beside the host-clock mean above
    ps = sorted(per_step_ms)
    out["ms_per_step_median_events"] = round(ps[len(ps) // 2] if len(ps) % 2 else 0.5 * (ps[len(ps) // 2 - 1] + ps[len(ps) // 2]), 3)
    out["ms_per_step_min_max_events"] = [round(ps[0], 3), round(ps[-1], 3)]
    from ssl_cr_histo_amd import _lib as _L
    import hashlib
    with open(_L.LIB_PATH, "rb") as f:
        so_sha = hashlib.sha256(f.read()).hexdigest()[:16]
    out["library"] = {"path": os.path.relpath(_L.LIB_PATH, ROOT), "so_mtime": int(os.path.getmtime(_L.LIB_PATH)), "so_sha16": so_sha,
                      "build_mode": "in-tree hipcc --offload-arch=gfx950 (ssl_cr_histo_amd/build.py), loaded through ctypes; no JIT, no fallback"}

    # roofline leg: same steps again with every conv launch bracketed by HIP events on its own stream.  EVERY rank runs these steps
    # (rank 0 stepping alone would wait for its peers for ever, and they for it at the final barrier); only rank 0 reads the table.
    rows = []
    nprof = max(2, min(5, args.steps))
    if not args.no_roofline:
        eng.profile(True)
        for _ in range(nprof):
            step()
        barrier()
        if rank == 0:
            rows = eng.profile_table()      # per kernel template instance, sorted by total time
        eng.profile(False)
    # the workload (modules, optimizer state, the engine nets' activations) goes into a holder that rank 0 empties before the side
    # configurations run in their child processes: they are measured on an otherwise empty device, as in round 4
    work = [step, keep]
    del step, keep
    if rank == 0:
        rank0_only_legs(args, world, out, rows, nprof, ms_per_step, work)
        out["parity"] = parity_record(out)
        print(json.dumps(out), flush=True)
    work.clear()
    barrier()                               # the final barrier of the job: every rank arrives here, whatever rank 0 did on its own


def rank0_only_legs(args, world, out, rows, nprof, ms_per_step, work):
    """Everything only rank 0 does.  NOTHING here may step the engine or call a collective when world > 1: the legs that run the
    workload again (counter passes, CPU baseline, side configurations) are child processes and are world == 1 only."""
    if not args.no_roofline:
        peak = 157.3 if args.dtype == "fp32" else 2500.0
        hbm_rows = [r for r in rows if r["flops"] == 0]
        rows = [r for r in rows if r["flops"] > 0]
        # counters: measured now by child rocprofv3 --pmc passes over this workload; the committed file is only the fallback
        pmc, pmc_src, pmc_measured, pmc_note = {}, None, False, "skipped (--no-pmc)"
        if not args.no_pmc and world == 1:
            doc, pmc_note = pmc_in_run(args)
            if doc:
                pmc = {e["kernel"]: e for e in doc["kernels"]}
                pmc_src, pmc_measured = f"rocprofv3 --pmc child passes of this run ({doc['wall_s']} s)", True
        if not pmc and os.path.exists(os.path.join(ROOT, PMC_FILE)) and args.workload == "ssl_cr" and args.dtype == "bf16":
            pmc = {e["kernel"]: e for e in json.load(open(os.path.join(ROOT, PMC_FILE)))["kernels"]}
            pmc_src = PMC_FILE
        out["pmc_status"] = {"measured_in_run": pmc_measured, "source": pmc_src, "note": pmc_note}

        def pmc_of(name):
            for k, e in pmc.items():
                if name.startswith(k) or k.startswith(name):
                    return e
            return None

        def roof(d):
            e = pmc_of(d["name"])
            common = {"kernel": d["name"], "launches": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
                      "algorithmic_mb_per_launch": round(d["bytes"] / d["launches"] / 1e6, 2),
                      "traffic": round(e["traffic_bytes_per_launch"]) if e and e.get("traffic_bytes_per_launch") else None,
                      "time_share_of_step": round(d["ms"] / (ms_per_step * nprof), 3)}
            if e:
                # rocprofv3 --pmc passes over this same command (tools/pmc_step.sh), averaged over the kernel's launches of a step
                common["pmc"] = {"measured_in_run": pmc_measured, "source": pmc_src, "mfma_busy": e.get("mfma_busy"),
                                 "valu_per_mfma": e.get("valu_per_mfma"),
                                 "traffic_over_algorithmic": round(e["traffic_bytes_per_launch"] / (d["bytes"] / d["launches"]), 3)
                                 if e.get("traffic_bytes_per_launch") else None}
            if d["flops"] > 0:
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                pk = 5000.0 if "fp8" in d["name"] else peak          # e4m3 kernels are priced against the 5 PF dense fp8 peak
                return dict(bound="mfma", achieved=round(ach, 2), peak=pk, unit="TFLOP/s", frac=round(ach / pk, 4),
                            algorithmic_gflop_per_launch=round(d["flops"] / d["launches"] / 1e9, 3), **common)
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            return dict(bound="hbm", achieved=round(ach, 1), peak=8000.0, unit="GB/s", frac=round(ach / 8000.0, 4), **common)
        if hbm_rows:
            out["roofline_hbm"] = roof(hbm_rows[0])       # the largest HBM-bound kernel (BatchNorm backward apply)
        if rows:
            d = rows[0]                       # the dominant MFMA kernel of the step
            if args.dtype == "fp8":           # config 5: the roofline object is the fp8 kernel's (largest fp8 row), peak 5 PF
                d = next((r for r in rows if "fp8" in r["name"]), d)
            out["roofline"] = roof(d)
            pc = out.get("power_clock") or {}
            if pc.get("sclk_mhz_mean"):
                # the spec peak is quoted at the 2.4 GHz boost clock; under the socket power limit the step runs below it (power_clock):
                # `frac` stays achieved / spec peak, this is the same achieved against the peak at the clock the part actually held
                f = pc["sclk_mhz_mean"] / 2400.0
                out["roofline"]["sclk_mhz_mean_of_step"] = pc["sclk_mhz_mean"]
                out["roofline"]["frac_of_peak_at_measured_sclk"] = round(out["roofline"]["achieved"] / (out["roofline"]["peak"] * f), 4)
            if hbm_rows and hbm_rows[0]["ms"] > d["ms"]:
                out["roofline_note"] = ("by total time the HBM-bound bn_bwd_apply kernel edges out the largest conv kernel; both "
                                        "roofs are reported (roofline = MFMA kernel, roofline_hbm = that kernel)")
            out["conv_kernels"] = []
            for r in rows:
                e = pmc_of(r["name"])
                out["conv_kernels"].append({"kernel": r["name"], "launches_per_step": r["launches"] // nprof,
                                            "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                                            "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1),
                                            "ms_per_step": round(r["ms"] / nprof, 3),
                                            "mfma_busy": e.get("mfma_busy") if e else None})
            tot_ms = sum(r["ms"] for r in rows)
            tot_fl = sum(r["flops"] for r in rows)
            out["conv_all"] = {"tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 1), "frac_of_peak": round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4),
                               "time_share_of_step": round(tot_ms / (ms_per_step * nprof), 3)}
            if pmc:
                busy = [(r["ms"], pmc_of(r["name"])) for r in rows]
                busy = [(m, e["mfma_busy"]) for m, e in busy if e and e.get("mfma_busy") is not None]
                if busy:
                    out["conv_all"]["mfma_busy_time_weighted"] = round(sum(m * b for m, b in busy) / sum(m for m, _ in busy), 4)
                    out["conv_all"]["mfma_busy_source"] = pmc_src
                    out["conv_all"]["mfma_busy_measured_in_run"] = pmc_measured
                    out["mfma_util_pct"] = round(100.0 * out["conv_all"]["mfma_busy_time_weighted"], 1)    # BASELINE metric's "MFMA util %"
                tr = [e.get("traffic_bytes_per_launch", 0) * e["dispatches_measured"] for e in pmc.values() if e.get("traffic_bytes_per_launch")]
                if tr and pmc_measured:
                    out["hbm_traffic_gb_per_step"] = round(sum(tr) / 4 / 1e9, 2)      # the child runs 2 warm-up + 2 timed steps

    if world == 1 and not args.no_cpu_baseline and args.workload == "ssl_cr":
        # the CPU baseline (host cores only, a child process with the GPU hidden) runs ALONE: after the roofline leg and the counter
        # passes, before the side legs -- nothing else of this command is running, so the baseline of record is not taken on a
        # host that is also driving the GPU from busy threads (round 3 overlapped them to save ~35 s of wall time)
        load0 = os.getloadavg()[0]
        out["cpu_baseline"] = cpu_baseline(args)
        if out["cpu_baseline"]:
            out["cpu_baseline"]["host"]["loadavg_1min_before"] = round(load0, 2)
            out["cpu_baseline"]["host"]["concurrent_with_gpu_legs"] = False
    if world == 1 and not args.no_also and args.workload == "ssl_cr" and args.dtype == "bf16":
        # the other BASELINE.json configurations on the same box (short runs: they are records, not the headline)
        import gc
        work.clear()                    # world == 1: nothing steps this workload again
        gc.collect()
        torch.cuda.empty_cache()
        out["also"] = also_records(args)


if __name__ == "__main__":
    main()
