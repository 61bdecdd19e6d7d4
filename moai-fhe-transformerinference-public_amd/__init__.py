"""MI355X-native RNS-CKKS evaluator hot path for MOAI (package `moai-fhe-transformerinference-public_amd`).

Everything here is a thin host binding over the C ABI of `libmoai_hip.so` (include/moai_hip.h), which
holds the hand-written gfx950 kernels.  There is no CPU fallback: importing `hip` raises if the
shared library has not been built (python __graft_entry__.py build, or make -C csrc).
"""
from . import hip, shard  # noqa: F401
from .hip import Context, DeviceBuffer, MoaiError, lib_path  # noqa: F401
