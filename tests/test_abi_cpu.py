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
    assert lib.sslcr_version() == 1
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
