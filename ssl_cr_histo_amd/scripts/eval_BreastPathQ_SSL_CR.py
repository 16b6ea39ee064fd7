"""engine-backed train()/validate() of the reference's eval_BreastPathQ_SSL_CR.py (see ssl_cr_histo_amd/steps.py)."""
from ..steps import bpq_cr_train as train  # noqa: F401
from ..steps import bpq_cr_validate as validate  # noqa: F401
from ..steps import teacher_refresh  # noqa: F401
