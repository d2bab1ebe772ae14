"""Truth tables of the 15 gate ids (kept in the package: tfhe_amd/truth.py)."""
from tfhe_amd.truth import TRUTH  # noqa: F401
