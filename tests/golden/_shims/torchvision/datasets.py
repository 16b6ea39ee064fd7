"""stub: the reference imports the name only on paths the goldens do not exercise."""
class _Any:
    def __getattr__(self, k):
        return _Any()
    def __call__(self, *a, **k):
        return _Any()
import sys
sys.modules[__name__].__class__ = type("M", (type(sys),), {"__getattr__": lambda self, k: _Any()})
