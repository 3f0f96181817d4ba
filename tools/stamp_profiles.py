#!/usr/bin/env python3
"""Stamp profiles/{traffic,valu}_<workload>.json with what bench.py needs to trust them:
  kernel_source_sha16  hash of the kernel sources the counters were measured on (bench.py prints null for `traffic` and
                       `roofline_valu` when it differs from the sources of the library it is timing);
  cheap_share          (valu files) share of plain 32-bit VOP1/VOP2 instructions in the dominant kernel's VALU stream, from the
                       ISA of the same sources: bench.py prices those at 2 issue cycles and everything else at 4.
Run here (needs the compiler, not a GPU) right after copying fresh counter summaries into profiles/.
usage: tools/stamp_profiles.py [workload...]   (default: every file present)"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import isa_budget  # noqa: E402

# dominant kernel(s) per workload: (symbol, main loop only?, relative weight = threads x launches of that kernel per unit)
DOMINANT = {
    "c3": [("_Z22relin_rows4_mac_kernelILb1EEv9RelinArgs", True, 1)],
    "c5": [("_Z22relin_rows4_mac_kernelILb1EEv9RelinArgs", True, 1)],
    "ntt": [("_Z15ntt_cols_kernelILb1ELb0ELi8EEv7NttArgs", False, 16), ("_Z20ntt_rows8_fwd_kernelILb1ELi0EEv7NttArgs", False, 32)],
    "c2": [("_Z22ntt_onepass_fwd_kernelILb1ELi6EEv7NttArgsjj", True, 1)],
    "c4": [("_Z20ntt_rows8_fwd_kernelILb1ELi4EEv7NttArgs", False, 1)],
}


def cheap_share(workload):
    num = den = 0.0
    for sym, loop, weight in DOMINANT[workload]:
        lines = isa_budget.body_of(sym)
        if loop:
            a, b = isa_budget.main_loop(lines)
            lines = lines[a:b + 1]
        valu, mad, cheap, *_ = isa_budget.classes(isa_budget.count(lines))
        num += weight * cheap
        den += weight * valu
    return num / den


want = sys.argv[1:]
sha = bench.kernel_source_sha16()
for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "valu_*.json"))):
    kind, workload = os.path.basename(path)[:-5].split("_", 1)
    if want and workload not in want:
        continue
    j = json.load(open(path))
    j["kernel_source_sha16"] = sha
    if kind == "valu" and workload in DOMINANT:
        j["cheap_share"] = round(cheap_share(workload), 4)
        j["cheap_share_note"] = "plain 32-bit VOP1/VOP2 (v_mov/not/add/sub/and/or/xor) share of the dominant kernel's VALU instructions, static ISA (tools/isa_budget.py)"
    json.dump(j, open(path, "w"), indent=1)
    print(os.path.relpath(path, ROOT), sha, j.get("cheap_share"))
