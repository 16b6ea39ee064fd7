"""engine-backed test() of the reference's test_Camelyon16.py (WSI tile classification -> probability map;
see ssl_cr_histo_amd/steps.py:camelyon16_test)."""
from ..steps import camelyon16_test as test  # noqa: F401
