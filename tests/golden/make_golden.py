#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.json from the CPU oracle (the reference ships no vectors and cannot be built here,
see DESIGN.md §2; these vectors pin THIS repository's arithmetic contract so that later changes to the oracle or the
kernels are caught).  Per case: parameters, seeds, SHA-256 of every output array and 16 sampled (limb, index, value)
triples.  Inputs are regenerated from the seeds by tests/golden_cases.py.   usage: python tests/golden/make_golden.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_cases import CASES, digest_outputs, run_case_oracle  # noqa: E402

out = {}
for name, case in CASES.items():
    out[name] = {"params": case, "outputs": digest_outputs(run_case_oracle(case))}
path = os.path.join(ROOT, "tests", "golden", "golden_v1.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path, len(out), "cases")
