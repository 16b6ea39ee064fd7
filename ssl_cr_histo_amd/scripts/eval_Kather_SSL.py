"""engine-backed train()/validate() of the reference's eval_Kather_SSL.py (see ssl_cr_histo_amd/steps.py)."""
from ..steps import kather_sup_train as train  # noqa: F401
from ..steps import kather_sup_validate as validate  # noqa: F401
