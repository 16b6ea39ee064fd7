"""Shared helpers for the oracle- and engine-side golden comparisons."""
import os
from collections import OrderedDict

import numpy as np
import torch

from oracle import cases as C
from oracle import model as OM

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"), allow_pickle=False)


def oracle_state(kind_cls, classes, rand_stats, seed=C.PARAM_SEED):
    """(params, buffers) dicts for net + classifier, seeded exactly like make_golden.build()."""
    sd = OM.init_state(seed, OM.net_param_specs(), random_running_stats=rand_stats)
    csd = OM.init_state(seed + 1, OM.classifier_param_specs(kind_cls, classes))
    p_net, b_net = OM.split_state(sd)
    p_cls, _ = OM.split_state(csd)
    return p_net, b_net, p_cls


def merged(p_net, p_cls):
    p = OrderedDict(p_net)
    p.update(p_cls)
    return p


def snapshot_dict(p, b):
    """name -> tensor, the oracle's analogue of state_dict()."""
    d = OrderedDict()
    for k, v in p.items():
        d[k] = v.detach()
    for k, v in b.items():
        d[k] = v
    return d


def check_snapshot(gold, prefix, state, rtol, atol_l2=1e-6):
    """compare the golden post-step snapshot (norms, sums, full tensors, nbt) with ``state``."""
    names = [str(n) for n in gold[f"{prefix}/names"]]
    l2 = gold[f"{prefix}/l2"]
    sm = gold[f"{prefix}/sum"]
    for i, k in enumerate(names):
        d = state[k].double()
        got = float(d.norm())
        assert abs(got - l2[i]) <= rtol * abs(l2[i]) + atol_l2, (k, got, l2[i])
        # sums cancel heavily; scale their tolerance by the tensor's L1 mass
        tol = rtol * float(d.abs().sum()) + 1e-6
        assert abs(float(d.sum()) - sm[i]) <= tol, (k, float(d.sum()), sm[i])
    for key in gold.files:
        if key.startswith(f"{prefix}/t/"):
            k = key[len(prefix) + 3:]
            want = torch.from_numpy(gold[key])
            if "[" in k:
                base, sl = k.split("[", 1)
                got = eval("state[base][" + sl)
            else:
                got = state[k]
            scale = float(want.abs().max()) + 1e-12
            err = float((got.float() - want).abs().max())
            assert err <= rtol * scale + 1e-7, (k, err, scale)
        elif key.startswith(f"{prefix}/nbt/"):
            k = key[len(prefix) + 5:]
            if k in state:
                assert int(state[k]) == int(gold[key]), (k, int(state[k]), int(gold[key]))


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
