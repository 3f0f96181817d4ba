"""bench.py: the algorithmic-byte figures are SURVEY.md 8(d)'s, the prime table is the oracle's, the JSON line carries the
contract's keys (GPU), and there is no CPU fallback."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_are_the_surveys():
    assert bench.algorithmic_bytes("c3", 1 << 16, 30, 30) == 1_037_041_664      # 1978 limbs x 512 KiB
    assert bench.algorithmic_bytes("c5", 1 << 15, 16, 1) == 158_859_264         # 606 limbs x 256 KiB
    assert bench.algorithmic_bytes("c4", 1 << 16, 30, 30) == 125_829_120        # 240 limbs x 512 KiB
    assert bench.algorithmic_bytes("ntt", 1 << 16, 8, 0) == 1_048_576           # 16 N
    assert bench.algorithmic_bytes("c2", 1 << 14, 8, 0) == 262_144


def test_prime_table_is_the_oracles():
    from oracle import oracle as orc
    for logN, count in ((10, 5), (16, 8)):
        pr, ps = bench.gen_primes_host(logN, count)
        opr, ops = orc.gen_primes(logN, count)
        assert np.array_equal(pr, opr) and np.array_equal(ps, ops)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,batch", [("c2", 64), ("c5", 2), ("dec", 2)])
def test_bench_line_carries_the_contract(workload, batch):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--batch", str(batch), "--steps", "2",
                        "--warmup", "1", "--prewarm", "0", "--cpu-budget", "0.5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "u64" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and j["value"] > 0
