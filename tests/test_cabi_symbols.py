"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/fhe_hip.h declares,
and refuses to create a context without a GPU (no CPU fallback).  No compute calls here."""
import os
import re

import numpy as np
import pytest

import fhe_amd
from fhe_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fhe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fhe_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = fhe_amd.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(hip.EXPORTED_SYMBOLS) == declared


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from oracle import oracle as orc
    pr, ps = orc.gen_primes(13, 4)
    with pytest.raises(fhe_amd.FheHipError) as ei:
        fhe_amd.HipContext(13, 2, 2, pr, ps)
    assert ei.value.code == -5


def test_product_path_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under fhe_amd/ or include/ may import or link it."""
    for base in ("fhe_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                    text = open(os.path.join(dp, f), errors="ignore").read()
                    assert "oracle" not in text.lower(), os.path.join(dp, f)
