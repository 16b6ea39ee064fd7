"""Modules named after the reference scripts, each exporting that script's ``train`` / ``validate`` (same signatures,
same return tuples) backed by the engine -- ``import ssl_cr_histo_amd.scripts.eval_BreastPathQ_SSL_CR as m; m.train(...)``
is the one-line swap for the reference's module-level functions.

There is no file per script: the table below names, for every reference script with a hot-path step function, the functions of
``ssl_cr_histo_amd.steps`` that replace it, and a meta-path finder serves ``ssl_cr_histo_amd.scripts.<script>`` as a module
holding exactly those names.  ``steps`` (and with it torch and the engine) is imported when such a module is first imported,
not with this package."""
import importlib
import importlib.abc
import importlib.machinery
import sys

# reference script -> {exported name: function in ssl_cr_histo_amd.steps}; file:line of the function each one replaces is in steps.py
SCRIPTS = {
    "eval_BreastPathQ_SSL_CR": {"train": "bpq_cr_train", "validate": "bpq_cr_validate", "teacher_refresh": "teacher_refresh"},
    "eval_Camelyon_SSL_CR": {"train": "cam_cr_train", "validate": "cam_cr_validate", "teacher_refresh": "teacher_refresh"},
    "eval_Kather_SSL_CR": {"train": "kather_cr_train", "validate": "kather_cr_validate", "teacher_refresh": "teacher_refresh"},
    "eval_BreastPathQ_SSL": {"train": "bpq_sup_train", "validate": "bpq_cr_validate", "teacher_refresh": "teacher_refresh"},
    "eval_Camelyon_SSL": {"train": "cam_sup_train", "validate": "cam_cr_validate", "teacher_refresh": "teacher_refresh"},
    "eval_Kather_SSL": {"train": "kather_sup_train", "validate": "kather_sup_validate", "teacher_refresh": "teacher_refresh"},
    "pretrain_BreastPathQ": {"train": "rsp_train", "validate": "rsp_validate", "teacher_refresh": "teacher_refresh"},
    "pretrain_Camelyon16": {"train": "rsp_train", "validate": "rsp_validate", "teacher_refresh": "teacher_refresh"},
    "pretrain_RSP": {"train": "rsp_train", "validate": "rsp_validate", "teacher_refresh": "teacher_refresh"},
    "test_Camelyon16": {"test": "camelyon16_test"},
}
__all__ = sorted(SCRIPTS)


class _ScriptFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        pkg, _, name = fullname.rpartition(".")
        if pkg == __name__ and name in SCRIPTS:
            # origin = this file: the synthesised modules have a __file__ / __spec__.origin for tracebacks, inspect and reload
            spec = importlib.machinery.ModuleSpec(fullname, self, origin=__file__)
            spec.has_location = True
            return spec
        return None

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        name = module.__name__.rpartition(".")[2]
        steps = importlib.import_module(__name__.rpartition(".")[0] + ".steps")
        module.__doc__ = f"engine-backed step functions of the reference's {name}.py (ssl_cr_histo_amd/steps.py)"
        for public, fn in SCRIPTS[name].items():
            setattr(module, public, getattr(steps, fn))


if not any(isinstance(f, _ScriptFinder) for f in sys.meta_path):
    sys.meta_path.append(_ScriptFinder())


def __dir__():                   # pkgutil / dir() see the script modules although there is no file per script
    return sorted(list(globals()) + list(SCRIPTS))


def __getattr__(name):          # `from ssl_cr_histo_amd.scripts import test_Camelyon16`
    if name in SCRIPTS:
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
