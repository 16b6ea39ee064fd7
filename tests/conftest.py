import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- order of the GPU suite (the driver runs `pytest -m gpu -x`: one failure hides everything behind it).  Cheap and diagnostic
# first -- per-kernel parity, then the kernel part of the fp8 file, run-to-run determinism, the layer-wise backward replay, the engine
# goldens -- and the long, error-amplifying ones last: multi-iteration trajectories, virtual ranks, the config-5 oracle run (75 s of
# host time).  Within a rank the collection order is kept (stable sort).
_FILE_RANK = {"test_kernels_gpu.py": 0, "test_fp8_gpu.py": 1, "test_determinism_gpu.py": 2, "test_backward_replay_gpu.py": 3,
              "test_engine_gpu.py": 4, "test_engine_gpu2.py": 5}
_LATE = (("test_virtual_ranks_at_the_headline", 7), ("test_trajectory_vs_reference", 8), ("test_config5_per_gpu_shape_vs_oracle", 9))


def _rank(item):
    fn = os.path.basename(str(item.fspath))
    name = item.name
    for key, r in _LATE:
        if name.startswith(key):
            return r
    if name.startswith("test_virtual_ranks_equal"):           # world 2 with the engine goldens, world 4 / 8 late
        world = item.callspec.params.get("world", 2) if hasattr(item, "callspec") else 2
        return 5 if world <= 2 else 7
    if fn == "test_fp8_gpu.py" and not name.startswith(("test_pack_fp8", "test_conv_fp8")):
        return 6                                   # the engine-level fp8 step after the engine goldens
    return _FILE_RANK.get(fn, 6)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_rank)
