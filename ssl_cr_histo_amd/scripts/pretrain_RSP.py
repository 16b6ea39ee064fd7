"""engine-backed train()/validate() of the reference's pretrain_RSP.py (see ssl_cr_histo_amd/steps.py)."""
from ..steps import rsp_train as train  # noqa: F401
from ..steps import rsp_validate as validate  # noqa: F401
from ..steps import teacher_refresh  # noqa: F401
