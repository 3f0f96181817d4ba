#!/usr/bin/env python3
"""Rewrite the rows of BASELINE.md section 4's results table from profiles/<round>/bench_*.json (default round r02) (run from the repository root)."""
import json, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r02"
s = open("BASELINE.md").read()
cut = s.index("## 4. Results")
head, tail = s[:cut], s[cut:]
d = {wl: json.load(open("profiles/%s/bench_%s.json" % (RND, wl))) for wl in ["c3", "ntt", "c4", "c5", "c2", "rotm", "rotnn", "dec", "enc"]}
f = lambda x: format(round(x), ",").replace(",", " ")


def ratios(r):
    cb = r["cpu_baseline"]
    return "%s× / %s×" % (f(r["value"] / cb["value"]), f(r["value"] / cb["all_cores"]["value"]))


def row_replace(prefix, newrow):
    global tail
    a = tail.index(prefix)
    b = tail.index("\n", a)
    tail = tail[:a] + newrow + tail[b:]


def pct(r):
    return 100 * r["roofline_whole_op"]["frac_of_hbm_peak"]


r = d["ntt"]; cb = r["cpu_baseline"]
row_replace("| NTT N=2^16", "| NTT N=2^16, 2048 limbs/step (1 GiB, beyond the 256 MB Infinity Cache) | %.2f M limb-NTT/s (≈4 %% more at 512 limbs/step, which the Infinity Cache half absorbs) | %.0f %% | cols+rows pair %s GB/s (%.2f) | %.2f GB (alg 2.15 GB × 2 passes) | %s /s / %s /s | %s |" % (
    r["value"] / 1e6, pct(r), f(r["roofline"]["achieved"]), r["roofline"]["frac"], (r["roofline"]["traffic"] or 0) / 1e9, f(cb["value"]), f(cb["all_cores"]["value"]), ratios(r)))
r = d["c2"]; cb = r["cpu_baseline"]
row_replace("| C2 N=2^14", "| C2 N=2^14, 8 limbs × 1024 polys, fwd+inv | %.2f M limb-NTT/s (one-pass kernels; %.2f M with the pass pair) | %.0f %% | `ntt_onepass_fwd_kernel` %s GB/s (%.2f); VALU issue slots %.0f %% busy | %.2f GB (alg %.2f GB: one pass over HBM) | %s /s / %s /s | %s |" % (
    r["value"] / 1e6, json.load(open("profiles/%s/bench_c2_passpair.json" % RND))["value"] / 1e6, pct(r), f(r["roofline"]["achieved"]), r["roofline"]["frac"], 100 * r["roofline_valu"]["frac_pmc_pass"], (r["roofline"]["traffic"] or 0) / 1e9, r["roofline"]["alg_bytes_per_launch"] / 1e9, f(cb["value"]), f(cb["all_cores"]["value"]), ratios(r)))
r = d["c3"]; cb = r["cpu_baseline"]; rf = r["roofline"]
row_replace("| C3 N=2^16", "| C3 N=2^16, Li=30, batch 16 | %s op/s | %.0f %% | `relin_rows8p_mac_kernel` %s GB/s (%.2f, keys charged once per launch; %.2f with SURVEY's per-op keys); VALU issue slots %.0f %% busy | %.2f GB (alg %.2f GB) | %.2f op/s / %.1f op/s | %s |" % (
    f(r["value"]), pct(r), f(rf["achieved"]), rf["frac"], rf["per_op_keys"]["frac"], 100 * r["roofline_valu"]["frac_pmc_pass"], (rf["traffic"] or 0) / 1e9, rf["alg_bytes_per_launch"] / 1e9, cb["value"], cb["all_cores"]["value"], ratios(r)))
r = d["c4"]; cb = r["cpu_baseline"]; rf = r["roofline"]
row_replace("| C4 N=2^16", "| C4 N=2^16, Li=K=30, batch 64 | %s op/s | %.0f %% | `bconv_kernel` (mod-down of both halves) %s GB/s (%.2f); VALU issue slots %.0f %% busy | %.2f GB (alg %.2f GB) | %.2f op/s / %.1f op/s | %s |" % (
    f(r["value"]), pct(r), f(rf["achieved"]), rf["frac"], 100 * r["roofline_valu"]["frac_pmc_pass"], (rf["traffic"] or 0) / 1e9, rf["alg_bytes_per_launch"] / 1e9, cb["value"], cb["all_cores"]["value"], ratios(r)))
r = d["c5"]; cb = r["cpu_baseline"]; rf = r["roofline"]
row_replace("| C5 N=2^15", "| C5 N=2^15, Li=16, batch 64, 1 GPU | %s op/s | %.0f %% | `relin_rows8p_mac_kernel` %s GB/s (%.2f) | — | %.2f op/s / %.1f op/s | %s |" % (
    f(r["value"]), pct(r), f(rf["achieved"]), rf["frac"], cb["value"], cb["all_cores"]["value"], ratios(r)))
names = {"rotm": "§8(f)-1 MixedRotKey rotation, N=2^16, Li=30, batch 16",
         "rotnn": "§8(f)-2 power-basis rotation (`rotate_slots_non_ntt`), N=2^16, Li=K=30, batch 16",
         "dec": "§8(f)-3 decrypt inner loops, N=2^16, Li=30, batch 64", "enc": "§8(f)-3 encrypt inner loops, N=2^16, Li=30, batch 64"}
for wl, n in names.items():
    r = d[wl]; cb = r["cpu_baseline"]; rf = r["roofline"]
    row_replace("| " + n, "| %s | %s op/s | %.0f %% | whole pipeline %s GB/s (%.2f) | — | %.2f op/s / %.1f op/s | %s |" % (
        n, f(r["value"]), pct(r), f(rf["achieved"]), rf["frac"], cb["value"], cb["all_cores"]["value"], ratios(r)))
open("BASELINE.md", "w").write(head + tail)
