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
    assert rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "kernel_ms" in j and j["kernel_ms"], "per-kernel durations of the untimed second pass"
    cb = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and j["value"] > 0


def test_dominant_kernel_bytes_do_not_credit_keys_per_ciphertext():
    """c3: one relin_rows8p_mac_kernel launch over a batch of 16 fetches the 1800 key limbs once (9.52 GB by the committed
    counters), not once per ciphertext (16.4 GB+): `frac` is computed from the shared figure, the per-op one is a labelled extra."""
    shared, per_op = bench.dominant_kernel_bytes("c3", 1 << 16, 30, 1, 16)
    assert shared == 8 * (1 << 16) * (1800 + 16 * (900 + 120)) and per_op == 16 * 8 * (1 << 16) * (2700 + 120)
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_c3.json")))["hbm_bytes_per_launch"]
    assert abs(shared - t) / t < 0.05, "the algorithmic bytes of the launch are what the PMC counters saw, within 5 %"
    for w in ("c3", "ntt", "c4"):
        v = json.load(open(os.path.join(ROOT, "profiles", "valu_%s.json" % w)))
        assert v["insts_valu_per_launch"] > 0 and v["gui_active_per_xcd_per_launch"] > 0 and v["launch_ms_in_pmc_pass"] > 0
        # the VALU block is a lower bound on busy issue cycles over available SIMD cycles, from this one pass: <= 1 by construction
        need = v["insts_valu_per_launch"] * (bench.VALU_ISSUE_CYCLES * (1 - v["cheap_share"]) + bench.VALU_CHEAP_CYCLES * v["cheap_share"])
        assert 0.3 < need / (bench.N_SIMD * v["gui_active_per_xcd_per_launch"]) <= 1.0, w


def test_committed_counter_files_match_the_kernel_sources_or_are_ignored(monkeypatch):
    """profiles/{traffic,valu}_*.json carry the hash of the kernel sources they were measured on (tools/stamp_profiles.py); bench.py
    uses them only for those sources and prints null otherwise.  The committed files must be current."""
    sha = bench.kernel_source_sha16()
    for kind in ("traffic", "valu"):
        for w in ("c3", "ntt", "c4", "c2"):
            j = json.load(open(os.path.join(ROOT, "profiles", "%s_%s.json" % (kind, w))))
            assert j.get("kernel_source_sha16") == sha, "%s_%s.json is stale: re-run tools/gpu_profile_round.sh + tools/stamp_profiles.py" % (kind, w)
            assert bench.read_profile_json(kind, w) is not None
    monkeypatch.setattr(bench, "kernel_source_sha16", lambda: "0" * 16)
    assert bench.read_profile_json("traffic", "c3") is None and bench.read_profile_json("valu", "c3") is None


def test_step_traffic_is_reported_against_the_operations_algorithmic_bytes():
    """c3: the counters' bytes over all kernels of a step against SURVEY 8(d)'s key-amortised bytes: the line says how much
    traffic the two-kernel design adds (the column-pass output written and read back)."""
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_c3.json")))
    N, Li, B = 1 << 16, 30, 16
    alg = B * 8 * N * (6 * Li - 2 + 2 * Li * Li / B)
    assert t["hbm_bytes_per_step"] > t["hbm_bytes_per_launch"]
    assert 5.0 < t["hbm_bytes_per_step"] / alg < 12.0
    assert abs(sum(t["hbm_bytes_per_step_by_kernel"].values()) - t["hbm_bytes_per_step"]) <= len(t["hbm_bytes_per_step_by_kernel"])


@pytest.mark.gpu
def test_default_line_is_honest_and_complete():
    """The driver's line (default workload and batch): `frac` never credits bytes the counters show were not moved, the VALU
    block is there and names what binds, limb-NTTs/sec is on the same line, and the timed region ran without per-launch events."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-budget", "0.5"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    rf, rv = j["roofline"], j["roofline_valu"]
    assert rf["traffic"] and rf["frac"] <= rf["traffic"] / (rf["avg_launch_ms"] * 1e-3) / 1e9 / rf["peak"] * 1.05 + 1e-3
    assert rf["per_op_keys"]["frac"] > rf["frac"]
    assert 0.5 < rv["frac"] <= 1.0 and rv["insts_valu_per_launch"] > 0 and rv["issue_cycles"] == {"plain_32bit_vop": 2, "other": 4}
    assert "min_bytes_of_this_design_per_launch" in rf and "alg_bytes_per_launch" not in rf
    assert rf["traffic_per_step"] > rf["traffic"] and rf["traffic_over_algorithmic"] > 5
    assert rf["bound"].startswith("valu") == (rv["frac"] > rf["frac"])
    n16 = j["ntt_n16"]
    assert n16["unit"] == "limb-NTT/s" and n16["limbs_per_step_per_gpu"] == 2048 and n16["value"] > 1e6
    assert n16["roofline"]["frac"] == pytest.approx(n16["roofline"]["achieved"] / 8000.0)
    assert "key_amortised" in j["roofline_whole_op"]
    assert j["cpu_baseline"]["kind"] == "port" and j["n_gpus"] == 1


def test_gpus_flag_must_agree_with_the_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in (r.stdout + r.stderr)
