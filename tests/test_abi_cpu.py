"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/sslcr.h declares;
the product path refuses to run without the MI355X (no CPU/PyTorch fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ssl_cr_histo_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    from ssl_cr_histo_amd import engine  # noqa: F401  registers engine signatures
    return _lib.lib()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sslcr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sslcr_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from ssl_cr_histo_amd import _lib
    names = header_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sslcr.h but not exported by libsslcr.so"
    missing = [n for n in names if n not in _lib.SIGNATURES]
    assert not missing, f"no ctypes signature for {missing}"
    extra = [n for n in _lib.SIGNATURES if n not in names]
    assert not extra, f"bound but not declared in the header: {extra}"


def test_version_and_error_string(lib):
    assert lib.sslcr_version() >= 4      # the ABI notes in include/sslcr.h refer to this number
    assert isinstance(lib.sslcr_last_error(), bytes)


def test_argument_validation_without_gpu(lib):
    """descriptor checks run before any device work: bad arguments return -1 with a message."""
    from ssl_cr_histo_amd import _lib as L
    d = L.ConvDesc()
    assert lib.sslcr_conv2d(0, d, None) == -1
    assert b"null tensor" in lib.sslcr_last_error()
    assert lib.sslcr_conv2d(7, d, None) == -1
    assert b"dtype" in lib.sslcr_last_error()
    assert lib.sslcr_net_forward(None, 0, None, None, None, 0, 1, 64, 64, None, None, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_path_fails_loudly_without_gpu(lib):
    from ssl_cr_histo_amd import _lib as L
    from ssl_cr_histo_amd import engine, kernels, net
    with pytest.raises(L.SslcrError):
        engine.Engine()
    x = torch.zeros(1, 8, 8, 64)
    w = torch.zeros(64, 3, 3, 64)
    with pytest.raises(L.SslcrError):
        kernels.conv2d(x, w, 1, 1)
    m = net.TripletNet_Finetune("resnet18")
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under ssl_cr_histo_amd/ may import it."""
    pkg = os.path.join(ROOT, "ssl_cr_histo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
    """every ctypes Structure of the Python layer against the C struct it mirrors: same size, same offset of every field
    (a small C program over include/sslcr.h, compiled with the host gcc).  Fields appended to a descriptor on one side only
    would otherwise show up as garbage pointers on the GPU."""
    import ctypes as C
    import shutil
    import subprocess
    from ssl_cr_histo_amd import _lib as L
    from ssl_cr_histo_amd import engine as E
    if not shutil.which("gcc"):
        pytest.skip("no host C compiler")
    pairs = {"sslcr_conv_desc": L.ConvDesc, "sslcr_wgrad_desc": L.WgradDesc, "sslcr_stem_desc": L.StemDesc,
             "sslcr_stem_wgrad_desc": L.StemWgradDesc, "sslcr_bn_finalize_desc": L.BnFinalizeDesc, "sslcr_bn_act_desc": L.BnActDesc,
             "sslcr_pool_fwd_desc": L.PoolFwdDesc, "sslcr_pool_bwd_desc": L.PoolBwdDesc, "sslcr_bn_bwd_desc": L.BnBwdDesc,
             "sslcr_loss_desc": L.LossDesc, "sslcr_tensor_desc": L.TensorDesc, "sslcr_opt_desc": L.OptDesc,
             "sslcr_pack_desc": L.PackDesc, "sslcr_weak_aug_desc": L.WeakAugDesc}
    for cname, pyname in (("sslcr_net_desc", "SslcrNetDesc"), ("sslcr_ssl_cr_desc", "SslCrDesc"), ("sslcr_sup_desc", "SupDesc")):
        assert hasattr(E, pyname), f"engine.py has no ctypes mirror of {cname}"
        pairs[cname] = getattr(E, pyname)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "sslcr.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} . %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr          # a field the header does not have fails here, by name
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    want = {}
    for ln in out.splitlines():
        cname, fname, v = ln.split()
        want[(cname, fname)] = int(v)
    for cname, cls in pairs.items():
        assert C.sizeof(cls) == want[(cname, ".")], f"sizeof {cname}: header {want[(cname, '.')]} vs ctypes {C.sizeof(cls)}"
        for fname, _ in cls._fields_:
            assert getattr(cls, fname).offset == want[(cname, fname)], f"{cname}.{fname}"
