"""Golden-generation aid ONLY (never imported by the product or the tests).

The reference imports ``torchvision.models.resnet18`` (torchvision==0.8.1, requirements.txt:416,
un-vendored; call sites models/net.py:3,32,77) and this image has no torchvision.  This is a
minimal stand-in with the same module tree / parameter names so that the reference's own
``models/net.py`` and ``train()`` functions can be imported by ``make_golden.py``.  All weights are
overwritten from ``oracle.model.init_state`` before anything is recorded, so no init recipe is
implied here.
"""
from . import models, transforms, datasets  # noqa: F401
