"""ssl_cr_histo_amd -- MI355X (gfx950) native engine for the SSL_CR_Histo ResNet18 step path.

Import of this package never touches the GPU library; the first engine use loads libsslcr.so and raises if it is
missing (there is no PyTorch/CPU fallback for the compute).
"""
from . import net  # noqa: F401
from .net import Classifier, FinetuneResNet, TripletNet, TripletNet_Finetune  # noqa: F401
from .util import AverageMeter  # noqa: F401

__all__ = ["net", "Classifier", "FinetuneResNet", "TripletNet", "TripletNet_Finetune", "AverageMeter"]
