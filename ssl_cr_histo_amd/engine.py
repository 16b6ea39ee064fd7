"""Python face of the native engine (libsslcr.so): binds the nn.Module parameter containers of net.py to a
``sslcr_net`` and runs whole steps with one C call.  PyTorch is only the owner of device memory, streams and the
torch.distributed bootstrap here -- there is no torch-op fallback for any compute.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib as L
from .net import Classifier, FinetuneResNet, TripletNet, TripletNet_Finetune, unwrap

_DTYPES = {"fp32": 0, "f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "fp8": 2}     # fp8: bf16 engine + e4m3 forward convs (config 5)

SslcrNetDesc = type("SslcrNetDesc", (C.Structure,), {"_fields_": [
    ("params", C.POINTER(C.c_void_p)), ("nparams", C.c_int), ("bn_running_mean", C.POINTER(C.c_void_p)),
    ("bn_running_var", C.POINTER(C.c_void_p)), ("bn_num_batches_tracked", C.POINTER(C.c_void_p)),
    ("requires_grad", C.POINTER(C.c_uint8)), ("head_kind", C.c_int), ("num_classes", C.c_int), ("triplet", C.c_int)]})
SslCrDesc = type("SslCrDesc", (C.Structure,), {"_fields_": [
    ("kind", C.c_int), ("x_student", C.c_void_p), ("x_teacher", C.c_void_p), ("in_f32", C.c_int), ("nx", C.c_int),
    ("nu", C.c_int), ("H", C.c_int), ("W", C.c_int), ("target_f", C.c_void_p), ("target_i", C.c_void_p),
    ("lambda_u", C.c_float), ("nx_global", C.c_int), ("nu_global", C.c_int), ("feats", C.c_void_p), ("logits", C.c_void_p),
    ("logits_t", C.c_void_p), ("losses", C.c_void_p), ("backward", C.c_int), ("x_student2", C.c_void_p)]})
SupDesc = type("SupDesc", (C.Structure,), {"_fields_": [
    ("kind", C.c_int), ("x1", C.c_void_p), ("x2", C.c_void_p), ("x3", C.c_void_p), ("in_f32", C.c_int), ("n", C.c_int),
    ("H", C.c_int), ("W", C.c_int), ("target_f", C.c_void_p), ("target_i", C.c_void_p), ("n_global", C.c_int),
    ("feats", C.c_void_p), ("logits", C.c_void_p), ("losses", C.c_void_p), ("train", C.c_int), ("backward", C.c_int)]})

_ENGINE_SIGS = {
    "sslcr_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "sslcr_destroy": (C.c_int, [C.c_void_p]),
    "sslcr_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "sslcr_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "sslcr_profile_dump": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "sslcr_comm_unique_id": (C.c_int, [C.c_void_p]),
    "sslcr_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "sslcr_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sslcr_comm_all_reduce_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sslcr_vcomm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "sslcr_vcomm_destroy": (C.c_int, [C.c_void_p]),
    "sslcr_comm_init_virtual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "sslcr_set_bn_sync": (C.c_int, [C.c_void_p, C.c_int]),
    "sslcr_set_aux_stream": (C.c_int, [C.c_void_p, C.c_int]),
    "sslcr_set_wgrad_stream": (C.c_int, [C.c_void_p, C.c_int]),
    "sslcr_net_create": (C.c_int, [C.c_void_p, C.POINTER(SslcrNetDesc), C.POINTER(C.c_void_p)]),
    "sslcr_net_destroy": (C.c_int, [C.c_void_p]),
    "sslcr_net_set_requires_grad": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8)]),
    "sslcr_net_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "sslcr_net_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sslcr_net_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sslcr_net_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "sslcr_net_debug_tap": (C.c_int, [C.c_void_p, C.c_int]),
    "sslcr_net_segments_used": (C.c_int, [C.c_void_p]),
    "sslcr_net_debug_tensor": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.c_void_p]),
    "sslcr_net_optimizer_step": (C.c_int, [C.c_void_p, C.POINTER(L.OptDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.c_void_p]),
    "sslcr_net_lookahead": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_float, C.c_void_p]),
    "sslcr_net_ema_from": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "sslcr_step_ssl_cr": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(SslCrDesc), C.c_void_p]),
    "sslcr_step_supervised": (C.c_int, [C.c_void_p, C.POINTER(SupDesc), C.c_void_p]),
}
L.SIGNATURES.update(_ENGINE_SIGS)


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _version_sig(tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors)


class BoundNet:
    """one (model, classifier) pair registered with the engine."""

    def __init__(self, engine, model, classifier):
        self.engine = engine
        self.model = model
        self.classifier = classifier
        self.triplet = isinstance(model, TripletNet)
        if not isinstance(model, (TripletNet, TripletNet_Finetune)):
            raise TypeError("model must be ssl_cr_histo_amd.net.TripletNet or TripletNet_Finetune")
        if isinstance(classifier, FinetuneResNet):
            self.head_kind, self.ncls = 0, classifier.classifier[0].out_features
        elif isinstance(classifier, Classifier):
            self.head_kind, self.ncls = 1, classifier.classifier[2].out_features
        else:
            raise TypeError("classifier must be ssl_cr_histo_amd.net.FinetuneResNet or Classifier")
        self.params = [p for _, p in model.named_parameters()] + [p for _, p in classifier.named_parameters()]
        dev = engine.device
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise L.SslcrError(f"parameters must be contiguous fp32 tensors on {dev} (got {p.device}, {p.dtype})")
        self.bns = model.model.bn_modules()
        self._keep = (_ptr_array(self.params), _ptr_array([b.running_mean for b in self.bns]),
                      _ptr_array([b.running_var for b in self.bns]), _ptr_array([b.num_batches_tracked for b in self.bns]))
        self._rg = self._rg_flags()
        d = SslcrNetDesc(self._keep[0], len(self.params), self._keep[1], self._keep[2], self._keep[3],
                         (C.c_uint8 * len(self.params))(*self._rg), self.head_kind, self.ncls, int(self.triplet))
        h = C.c_void_p()
        L.check(L.lib().sslcr_net_create(engine.handle, C.byref(d), C.byref(h)))
        self.handle = h
        self._sig = None
        self._finalizer = weakref.finalize(self, BoundNet._destroy, h)

    @staticmethod
    def _destroy(h):
        try:
            L.lib().sslcr_net_destroy(h)
        except Exception:
            pass

    def _rg_flags(self):
        return [int(p.requires_grad) for p in self.params]

    def _tensors(self):
        out = list(self.params)
        for b in self.bns:
            out += [b.running_mean, b.running_var]
        return out

    def still_valid(self):
        """False when a parameter/buffer was re-allocated (e.g. module.to(), load into new storage)."""
        cur = [p for _, p in self.model.named_parameters()] + [p for _, p in self.classifier.named_parameters()]
        return len(cur) == len(self.params) and all(a is b for a, b in zip(cur, self.params)) and \
            all(a.data_ptr() == k for a, k in zip(self.params, self._keep[0]))

    def sync(self):
        """propagate host-side changes (requires_grad flips, load_state_dict, in-place edits) to the engine."""
        rg = self._rg_flags()
        if rg != self._rg:
            self._rg = rg
            L.check(L.lib().sslcr_net_set_requires_grad(self.handle, (C.c_uint8 * len(rg))(*rg)))
        sig = _version_sig(self._tensors())
        if sig != self._sig:
            L.check(L.lib().sslcr_net_pack(self.handle, 3, L.stream_ptr()))
            self._sig = sig

    # ------------------------------------------------------------------ forward / backward building blocks
    def forward(self, xs, train):
        """xs: tuple of 1 (Finetune) or 3 (TripletNet) NCHW uint8|fp32 device tensors -> (feats [N,768], logits [N,C])."""
        self.sync()
        xs = [self.engine.as_input(x) for x in xs]
        N, _, H, W = xs[0].shape
        feats = torch.empty((N, 768), dtype=torch.float32, device=self.engine.device)
        logits = torch.empty((N, self.ncls), dtype=torch.float32, device=self.engine.device)
        p = [L.ptr(x) for x in xs] + [None] * (3 - len(xs))
        L.check(L.lib().sslcr_net_forward(self.handle, int(train), p[0], p[1], p[2], int(xs[0].dtype == torch.float32),
                                          N, H, W, L.ptr(feats), L.ptr(logits), L.stream_ptr()))
        self._live_inputs = xs          # the stem wgrad re-reads the inputs in backward
        if train:
            self._note_buffers_changed()
        return feats, logits

    def backward(self, dlogits):
        L.check(L.lib().sslcr_net_backward(self.handle, L.ptr(dlogits.contiguous()), L.stream_ptr()))

    def grad(self, index):
        """gradient of parameter `index` (named_parameters order) in PyTorch layout -- for tests/inspection."""
        out = torch.empty_like(self.params[index])
        L.check(L.lib().sslcr_net_grad(self.handle, index, L.ptr(out), L.stream_ptr()))
        return out

    @property
    def segments_used(self):
        """True when the last train-mode forward ran the TripletNet branches as segments of one launch per layer."""
        return bool(L.lib().sslcr_net_segments_used(self.handle))

    def debug_tap(self, on):
        """keep copies of every block's transient gradient tensors in backward (layer-wise parity test; see include/sslcr.h)."""
        L.check(L.lib().sslcr_net_debug_tap(self.handle, int(bool(on))))

    def debug_tensor(self, block, kind):
        """-> (tensor in the engine's storage dtype, flags); kinds as in include/sslcr.h:sslcr_net_debug_tensor."""
        dims, flags = (C.c_int * 4)(), C.c_int()
        L.check(L.lib().sslcr_net_debug_tensor(self.handle, block, kind, None, 0, dims, C.byref(flags), L.stream_ptr()))
        shape = [d for d in dims]
        dt = torch.float32 if kind >= 11 or self.engine.dtype == 0 else torch.bfloat16
        out = torch.empty(shape if kind < 11 else shape[:1], dtype=dt, device=self.engine.device)
        L.check(L.lib().sslcr_net_debug_tensor(self.handle, block, kind, L.ptr(out), out.numel() * out.element_size(), dims,
                                               C.byref(flags), L.stream_ptr()))
        return out, flags.value

    def _note_buffers_changed(self):
        # the engine updated running stats / will update params through raw pointers: keep our signature in step so
        # that only *external* edits trigger a repack (the engine invalidates its own packs itself)
        self._sig = _version_sig(self._tensors())

    # ------------------------------------------------------------------ optimizer (state lives in the torch optimizer)
    def optimizer_step(self, optimizer):
        inner = getattr(optimizer, "optimizer", optimizer)           # Lookahead wraps the real optimizer
        if len(inner.param_groups) != 1:
            raise NotImplementedError("the reference builds a single param group; multiple groups are not supported")
        g = inner.param_groups[0]
        ids = {id(p) for p in g["params"]}
        mine = {id(p) for p in self.params if p.requires_grad}
        if ids != mine:
            raise L.SslcrError("optimizer parameters must be exactly the requires_grad parameters of (model, classifier) "
                               "-- the reference builds it with filter(lambda p: p.requires_grad, ...)")
        s1, s2 = [None] * len(self.params), [None] * len(self.params)
        if isinstance(inner, torch.optim.Adam):
            if g.get("amsgrad", False) or g.get("maximize", False):
                raise NotImplementedError("amsgrad/maximize are not used by the reference")
            step = None
            for i, p in enumerate(self.params):
                if not p.requires_grad:
                    continue
                st = inner.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                step = float(st["step"])
                s1[i], s2[i] = st["exp_avg"], st["exp_avg_sq"]
            b1, b2 = g["betas"]
            o = L.OptDesc(0, g["lr"], b1, b2, g["eps"], g["weight_decay"], 0.0, 1 - b1 ** step, 1 - b2 ** step, 0, 1.0)
        elif isinstance(inner, torch.optim.SGD):
            if not g.get("nesterov", False) or g.get("dampening", 0) != 0:
                raise NotImplementedError("the reference uses SGD(momentum, nesterov=True, dampening=0)")
            first = False
            for i, p in enumerate(self.params):
                if not p.requires_grad:
                    continue
                st = inner.state[p]
                if st.get("momentum_buffer", None) is None:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    first = True
                s1[i] = st["momentum_buffer"]
            o = L.OptDesc(1, g["lr"], 0.0, 0.0, 0.0, g["weight_decay"], g["momentum"], 1.0, 1.0, int(first), 1.0)
        else:
            raise NotImplementedError(f"optimizer {type(inner).__name__}: the reference uses Adam and SGD-Nesterov")
        self._opt_keep = (s1, s2, _ptr_array(s1), _ptr_array(s2))
        L.check(L.lib().sslcr_net_optimizer_step(self.handle, C.byref(o), self._opt_keep[2], self._opt_keep[3], L.stream_ptr()))
        self._note_buffers_changed()

    def lookahead(self, cached, alpha):
        arr = _ptr_array(cached)
        L.check(L.lib().sslcr_net_lookahead(self.handle, arr, float(alpha), L.stream_ptr()))
        self._note_buffers_changed()

    def ema_from(self, student, decay):
        L.check(L.lib().sslcr_net_ema_from(self.handle, student.handle, float(decay), L.stream_ptr()))
        self._note_buffers_changed()


class Engine:
    """one per process and device (one process per GPU)."""

    def __init__(self, device=None, dtype=None):
        if not torch.cuda.is_available():
            raise L.SslcrError("no MI355X visible: the engine has no CPU path")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        dtype = dtype or os.environ.get("SSLCR_DTYPE", "bf16")
        if dtype not in _DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_DTYPES)}")
        self.dtype = _DTYPES[dtype]
        lib = L.lib()
        for name, (res, args) in _ENGINE_SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(lib.sslcr_create(C.byref(h), self.device.index, self.dtype))
        self.handle = h
        self._bound = {}
        self._dummy_cls = None
        self.rank, self.world = 0, 1

    # ------------------------------------------------------------------ distributed (one process per GPU, RCCL inside)
    def init_comm(self, rank, world, broadcast_bytes):
        """broadcast_bytes(bytes_or_None) -> bytes : host-side broadcast of the 128-byte RCCL id from rank 0."""
        if world == 1 and not os.environ.get("SSLCR_COMM_SELFTEST"):
            return
        idbuf = (C.c_char * 256)()
        if rank == 0:
            L.check(L.lib().sslcr_comm_unique_id(idbuf))
        raw = broadcast_bytes(bytes(idbuf) if rank == 0 else None)
        idbuf = (C.c_char * 256).from_buffer_copy(raw)
        L.check(L.lib().sslcr_comm_init(self.handle, idbuf, rank, world))
        self.rank, self.world = rank, world

    def init_comm_virtual(self, vcomm, rank, world):
        """join a VirtualComm (world contexts of one process on one device, see include/sslcr.h): test infrastructure that runs
        the engine's sharded code paths on a single-GPU box."""
        L.check(L.lib().sslcr_comm_init_virtual(self.handle, vcomm.handle, rank))
        self.rank, self.world = rank, world
        self._vcomm = vcomm                                        # keep it alive as long as this engine

    def comm_info(self):
        """(rank, world, transport) as the communicator itself reports them; transport: 'none' | 'rccl' | 'virtual'."""
        r, w, t = C.c_int(), C.c_int(), C.c_int()
        L.check(L.lib().sslcr_comm_info(self.handle, C.byref(r), C.byref(w), C.byref(t)))
        return r.value, w.value, ("none", "rccl", "virtual")[t.value]

    def set_bn_sync(self, on):
        """True (default): train-mode BatchNorm uses global-batch statistics across ranks; False: per-replica statistics
        like the reference's nn.DataParallel (eval_BreastPathQ_SSL_CR.py:474-477)."""
        L.check(L.lib().sslcr_set_bn_sync(self.handle, int(bool(on))))

    def set_aux_stream(self, on):
        """True: step_ssl_cr runs the teacher forward on a second stream next to the student forward (-2 % step time; per-kernel
        timings then include the sharing of the CUs).  Default off."""
        L.check(L.lib().sslcr_set_aux_stream(self.handle, int(bool(on))))

    def set_wgrad_stream(self, on):
        """True: backward's weight-gradient launches run on a second stream (bit-identical results; measured neutral on MI355X,
        off by default -- see include/sslcr.h)."""
        L.check(L.lib().sslcr_set_wgrad_stream(self.handle, int(bool(on))))

    def all_reduce_sum(self, t):
        """in-place SUM of a contiguous fp32 device tensor over the ranks of the engine's own communicator (RCCL or virtual), on
        the current stream; identity on one rank.  Logging path only (steps._Meters at print_freq / epoch end): the data-path
        collectives -- gradient buckets, BatchNorm sums -- are issued inside the step by the engine itself."""
        if self.world > 1:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise L.SslcrError("all_reduce_sum: contiguous fp32 tensor on the engine's device expected")
            L.check(L.lib().sslcr_comm_all_reduce_f32(self.handle, L.ptr(t), t.numel(), L.stream_ptr()))
        return t

    def copy_stream(self):
        """side stream for host-to-device prefetch of the next batch (steps._ahead); torch owns streams, the engine only keeps one"""
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(self.device)
        return self._copy_stream

    # ------------------------------------------------------------------ measurement
    def profile(self, enable):
        L.check(L.lib().sslcr_profile(self.handle, int(enable)))

    def profile_read(self, which):
        """{launches, ms, flops, bytes} of the event-bracketed conv launches (0 conv_igemm fwd+dgrad, 1 wgrad)."""
        out = (C.c_double * 4)()
        L.check(L.lib().sslcr_profile_read(self.handle, which, out))
        return dict(launches=int(out[0]), ms=out[1], flops=out[2], bytes=out[3])

    def profile_table(self):
        """per kernel instance: [dict(name, launches, ms, flops, bytes)], sorted by total time."""
        buf = C.create_string_buffer(1 << 16)
        L.check(L.lib().sslcr_profile_dump(self.handle, buf, len(buf)))
        rows = []
        for line in buf.value.decode().splitlines():
            name, n, ms, fl, by = line.rsplit("|", 4)
            rows.append(dict(name=name, launches=int(float(n)), ms=float(ms), flops=float(fl), bytes=float(by)))
        return sorted(rows, key=lambda r: -r["ms"])

    # ------------------------------------------------------------------ binding
    def as_input(self, x):
        if x.dtype not in (torch.uint8, torch.float32):
            x = x.float()
        if x.device != self.device:
            x = x.to(self.device, non_blocking=True)
        return x.contiguous()

    def bind(self, model, classifier):
        model = unwrap(model)
        classifier = unwrap(classifier) if classifier is not None else self._dummy(model)
        key = (id(model), id(classifier))
        b = self._bound.get(key)
        if b is not None and (b.model is not model or b.classifier is not classifier or not b.still_valid()):
            b = None
        if b is None:
            b = BoundNet(self, model, classifier)
            self._bound[key] = b
            # drop bindings whose modules were garbage collected / replaced (teacher = deepcopy(student) every epoch)
            for k in [k for k, v in self._bound.items() if v is not b and not v.still_valid()]:
                del self._bound[k]
        return b

    def _dummy(self, model):
        if self._dummy_cls is None:
            self._dummy_cls = FinetuneResNet(1).to(self.device)
        return self._dummy_cls

    # ------------------------------------------------------------------ fused steps
    def step_ssl_cr(self, teacher, student, kind, x, y, u_w, u_s, lambda_u, backward=True, nx_global=None, nu_global=None):
        """one consistency-training iteration (eval_BreastPathQ_SSL_CR.py:65-100 / eval_Camelyon_SSL_CR.py:94-121).
        x [nx,3,H,W], u_w/u_s [nu,3,H,W] uint8|fp32; y [nx] fp32 (kind 'mse') or int64 (kind 'ce').
        -> dict(losses [4] device tensor: loss, loss_x, loss_u, #correct ; feats ; logits ; logits_t).  With more than one rank the
        losses are this rank's share (scaled by 1/global-count; the SUM over ranks is the global value) -- no collective per step."""
        teacher.sync()
        student.sync()
        x, u_w, u_s = self.as_input(x), self.as_input(u_w), self.as_input(u_s)
        if x.dtype != u_s.dtype or x.dtype != u_w.dtype:
            x, u_w, u_s = x.float(), u_w.float(), u_s.float()
        # torch.cat((inputs_x, inputs_u_s)) of :82 is not materialised: the stem kernel reads the two segments in place
        nx, nu = x.shape[0], u_w.shape[0]
        if u_s.shape[0] != nu or x.shape[1:] != u_s.shape[1:]:
            raise ValueError("step_ssl_cr: u_w / u_s batch sizes or image shapes differ")
        _, _, H, W = x.shape
        dev = self.device
        feats = torch.empty((nx + nu, 768), dtype=torch.float32, device=dev)
        logits = torch.empty((nx + nu, student.ncls), dtype=torch.float32, device=dev)
        logits_t = torch.empty((nu, student.ncls), dtype=torch.float32, device=dev)
        losses = torch.empty(4, dtype=torch.float32, device=dev)
        k = {"mse": 0, "ce": 1}[kind]
        y = y.to(dev).contiguous()
        tf = y.float() if k == 0 else None
        ti = y.long() if k == 1 else None
        d = SslCrDesc(k, x.data_ptr(), u_w.data_ptr(), int(x.dtype == torch.float32), nx, nu, H, W,
                      None if tf is None else tf.data_ptr(), None if ti is None else ti.data_ptr(), float(lambda_u),
                      int(nx_global or nx * self.world), int(nu_global or nu * self.world), feats.data_ptr(),
                      logits.data_ptr(), logits_t.data_ptr(), losses.data_ptr(), int(backward), u_s.data_ptr())
        L.check(L.lib().sslcr_step_ssl_cr(teacher.handle, student.handle, C.byref(d), L.stream_ptr()))
        student._live_inputs = (x, u_s, u_w, tf, ti)
        student._note_buffers_changed()
        return dict(losses=losses, feats=feats, logits=logits, logits_t=logits_t)

    def step_supervised(self, net, kind, xs, y, train=True, backward=True, n_global=None):
        """student-only step: RSP pretraining (TripletNet, 3 inputs, 'ce'; pretrain_BreastPathQ.py:42-61), supervised
        fine-tuning (eval_Camelyon_SSL.py:52-98 'ce', eval_BreastPathQ_SSL.py:52-84 'mse') and every validate()."""
        net.sync()
        xs = [self.as_input(x) for x in xs]
        n, _, H, W = xs[0].shape
        dev = self.device
        feats = torch.empty((n, 768), dtype=torch.float32, device=dev)
        logits = torch.empty((n, net.ncls), dtype=torch.float32, device=dev)
        losses = torch.empty(4, dtype=torch.float32, device=dev)
        k = {"ce": 2, "mse": 3}[kind]
        y = y.to(dev).contiguous()
        tf = y.float() if k == 3 else None
        ti = y.long() if k == 2 else None
        p = [x.data_ptr() for x in xs] + [None] * (3 - len(xs))
        d = SupDesc(k, p[0], p[1], p[2], int(xs[0].dtype == torch.float32), n, H, W, None if tf is None else tf.data_ptr(),
                    None if ti is None else ti.data_ptr(), int(n_global or n * self.world), feats.data_ptr(),
                    logits.data_ptr(), losses.data_ptr(), int(train), int(backward and train))
        L.check(L.lib().sslcr_step_supervised(net.handle, C.byref(d), L.stream_ptr()))
        net._live_inputs = (xs, tf, ti)
        if train:
            net._note_buffers_changed()
        return dict(losses=losses, feats=feats, logits=logits)


class VirtualComm:
    """shared exchange object of `world` virtual ranks (see include/sslcr.h: sslcr_vcomm)."""

    def __init__(self, world):
        lib = L.lib()
        for name in ("sslcr_vcomm_create", "sslcr_vcomm_destroy", "sslcr_comm_init_virtual"):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = _ENGINE_SIGS[name]
        h = C.c_void_p()
        L.check(lib.sslcr_vcomm_create(C.byref(h), world))
        self.handle, self.world = h, world

    def close(self):
        if self.handle:
            L.lib().sslcr_vcomm_destroy(self.handle)
            self.handle = None


_engines = {}


def get_engine(device=None, dtype=None):
    """process-wide engine for `device` (created on first use; dtype from SSLCR_DTYPE, default bf16)."""
    dev = torch.device(device if device is not None else "cuda")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    e = _engines.get(idx)
    if e is None:
        e = Engine(torch.device("cuda", idx), dtype)
        _engines[idx] = e
    elif dtype is not None and _DTYPES[dtype] != e.dtype:
        raise L.SslcrError("an engine with a different dtype already exists on this device; use set_engine()")
    return e


def set_engine(engine):
    _engines[engine.device.index] = engine
    return engine
