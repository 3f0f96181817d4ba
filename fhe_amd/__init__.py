"""fhe_amd — MI355X (gfx950) backend for fHE's RNS-CKKS polynomial hot path.

Only what the path needs: csrc/ (HIP kernels + the C-ABI of include/fhe_hip.h), lib/ (the built
libfhe_hip.so) and hip.py (ctypes loader + thin wrappers that hand torch/numpy buffers to the C-ABI).
There is no CPU fallback: loading fails loudly when the library is missing, and context creation
fails when no GPU is present.
"""
from .hip import HipContext, FheHipError, load_library, LIB_PATH  # noqa: F401
