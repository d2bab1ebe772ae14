"""tfhe_amd - MI355X-native batched gate bootstrapping behind libtfhe's API.

Python here is plumbing over the C-ABI library (include/tfhe_hip.h); the product is the HIP code in csrc/.
"""
from .engine import Engine, Group, Gate, TfheHipError, PARAMS, PARAMS_80, ParamSet  # noqa: F401
from .client import Client, KeySet  # noqa: F401
from . import wire  # noqa: F401
