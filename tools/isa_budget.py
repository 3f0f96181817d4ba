#!/usr/bin/env python3
"""Static VALU instruction budget of the NTT / key-switch kernels against the ideal butterfly.
Compiles fhe_amd/csrc/fhe_hip.hip to gfx950 assembly and counts, per kernel (for the relinearisation kernel: per iteration of
its digit loop, located as the largest loop body), the vector-ALU instructions by class.  The ideal is 18 per butterfly
(6 v_mad_u64_u32 + 12 word operations) and 16 per key multiply-accumulate (11 + 5).
usage: tools/isa_budget.py [out.txt]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = os.path.join(tempfile.gettempdir(), "fhe_budget.s")
text = None


def compile_asm():
    global text
    if text is None:
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "--rocm-path=/opt/rocm", "-O3", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", "-S",
                               "--cuda-device-only", "-w", "-o", asm, os.path.join(ROOT, "fhe_amd", "csrc", "fhe_hip.hip")])
        text = open(asm).read()
    return text


def body_of(sym):
    compile_asm()
    m = re.search(r'^(%s):[^\n]*\n(.*?)^\.Lfunc_end' % re.escape(sym), text, re.S | re.M)
    return [l.strip() for l in m.group(2).split('\n')]


def count(lines):
    ops = collections.Counter()
    for l in lines:
        if not l or l.startswith(('.', ';')) or l.endswith(':'):  # directives, comments, labels
            continue
        ops[l.split()[0]] += 1
    return ops


def main_loop(lines):
    """(start, end) of the innermost loop with the most multiply-adds."""
    labels = {l.split(':')[0]: i for i, l in enumerate(lines) if l.startswith('.LBB') and ':' in l}   # (loop headers carry a trailing comment)
    spans = []
    for i, l in enumerate(lines):
        if l.startswith('s_cbranch') or l.startswith('s_branch'):
            tgt = l.split()[-1]
            if tgt in labels and labels[tgt] < i:
                spans.append((labels[tgt], i, sum(1 for x in lines[labels[tgt]:i] if x.startswith('v_mad_u64_u32'))))
    # the innermost loop (no other loop nested in it) with the most multiply-adds: the per-digit (or per-pair-of-digits) loop,
    # not an outer loop that also holds peeled first / last iterations
    inner = [s for s in spans if not any(o is not s and s[0] <= o[0] and o[1] <= s[1] for o in spans)]
    a, b, _ = max(inner, key=lambda s: s[2])
    return a, b


def classes(ops):
    valu = sum(v for k, v in ops.items() if k.startswith('v_'))
    mad = sum(v for k, v in ops.items() if k.startswith('v_mad_u64_u32'))
    cheap = sum(v for k, v in ops.items() if k.split('_e')[0] in ('v_mov_b32', 'v_not_b32', 'v_add_u32', 'v_sub_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32'))
    return valu, mad, cheap, ops.get('s_nop', 0), sum(v for k, v in ops.items() if k.startswith('ds_')), sum(v for k, v in ops.items() if k.startswith('global_'))


def cheap_share(sym, loop=False):
    """share of the plain 32-bit VOP1/VOP2 class among a kernel's VALU instructions (loop: of its main loop's)"""
    lines = body_of(sym)
    if loop:
        a, b = main_loop(lines)
        lines = lines[a:b + 1]
    valu, mad, cheap, nops, lds, vmem = classes(count(lines))
    return cheap / valu


def report(argv):
    K = [
        # symbol, label, butterflies per thread, key MACs / scaling multiplies per thread, canonical outputs per thread, loop?
        ("_Z15ntt_cols_kernelILb1ELb0ELi8EEv7NttArgs", "ntt_cols_kernel<SP,fwd,8>: stand-alone and relinearisation column pass", 64, 0, 0, False),
        ("_Z20ntt_rows8_fwd_kernelILb1ELi0EEv7NttArgs", "ntt_rows8_fwd_kernel<SP,0>: stand-alone forward row pass", 32, 0, 8, False),
        ("_Z20ntt_rows8_inv_kernelILb1ELb0ELb0EEv7NttArgs", "ntt_rows8_inv_kernel<SP>: inverse row pass", 32, 0, 0, False),
        ("_Z15ntt_cols_kernelILb1ELb1ELi8EEv7NttArgs", "ntt_cols_kernel<SP,inv,8>: inverse column pass, scaling folded into the last stage (8 extra multiplies)", 64, 8, 16, False),
        ("_Z23relin_rows8p_mac_kernelILb1EEv9RelinArgs", "relin_rows8p_mac_kernel<SP> (A/B, rounds 1-2): one digit of the j-loop (row pass + 16 Montgomery MACs)", 32, 16, 0, True),
        # round 3 default: two digits side by side, 4 elements per thread; a key MAC is 6 multiply-adds into carry-free sums plus 5
        # instructions per element and digit for the 3 x 22-bit operand split, (6 * 16 + 5 * 8) / 16 = 8.5 per MAC against 16
        ("_Z22relin_rows4_mac_kernelILb1EEv9RelinArgs", "relin_rows4_mac_kernel<SP>: one PAIR of digits of the j-loop (2 row passes of 4 elements + 16 unreduced MACs)", 32, 16, 0, True),
        # the rotation pipeline (C4) beside the column passes above
        ("_Z20ntt_rows8_fwd_kernelILb1ELi4EEv7NttArgs", "ntt_rows8_fwd_kernel<SP,4>: row pass + key product of the normal limbs + mod-down tail (2 multiplies per element)", 32, 16, 8, False),
        ("_Z24rot_special_rows8_kernelILb1EEv9RotSpArgs", "rot_special_rows8_kernel<SP>: forward row pass + 2 key products + 2 inverse row passes", 96, 16, 0, False),
    ]
    # the generic-prime instantiations (any odd p < 2^62: what a drop-in caller's own table runs): ideal = 23 per butterfly
    # (16-instruction word-serial Montgomery, modarith.h:mont_generic, + 7), 21 per key MAC / scaling multiply
    KG = [(sym.replace("ILb1E", "ILb0E", 1), label.replace("<SP", "<generic"), bf, macs, canon, loop) for sym, label, bf, macs, canon, loop in K]
    out = ["static VALU budget per thread, gfx950 (-O3); ideal = 18 per butterfly + 16 per key MAC / scaling multiply + 8 per canonical output",
           "%-82s %5s %5s %6s %6s %8s %7s %5s %6s %6s" % ("kernel", "bfly", "MACs", "VALU", "ideal", "VALU/bf", "excess", "mads", "cheap", "s_nop")]
    for sym, label, bf, macs, canon, loop in K:
        lines = body_of(sym)
        if loop:
            a, b = main_loop(lines)
            lines = lines[a:b + 1]
        valu, mad, cheap, nops, lds, vmem = classes(count(lines))
        ideal = 18 * bf + 16 * macs + 8 * canon
        out.append("%-82s %5d %5d %6d %6d %8.2f %6.1f%% %5d %6d %6d   (LDS %d, vector-memory %d)" %
                   (label, bf, macs, valu, ideal, (valu - 16 * macs - 8 * canon) / bf, 100.0 * (valu - ideal) / ideal, mad, cheap, nops, lds, vmem))
    out.append("generic primes: ideal = 23 per butterfly + 21 per key MAC / scaling multiply + 8 per canonical output")
    for sym, label, bf, macs, canon, loop in KG:
        lines = body_of(sym)
        if loop:
            a, b = main_loop(lines)
            lines = lines[a:b + 1]
        valu, mad, cheap, nops, lds, vmem = classes(count(lines))
        ideal = 23 * bf + 21 * macs + 8 * canon
        out.append("%-82s %5d %5d %6d %6d %8.2f %6.1f%% %5d %6d %6d   (LDS %d, vector-memory %d)" %
                   (label, bf, macs, valu, ideal, (valu - 21 * macs - 8 * canon) / bf, 100.0 * (valu - ideal) / ideal, mad, cheap, nops, lds, vmem))
    out.append("VALU/bf = (VALU - 16*MACs - 8*canonical outputs) / butterflies; cheap = plain 32-bit VOP1/VOP2 (v_mov/not/add/sub/and/or/xor): the only class that")
    out.append("issues faster than a multiply-add, and only back to back with another of its kind (instrate2.txt: mad + add32 = 8.8 cycles, mad + 2 add32 = 10.4).")
    res = "\n".join(out)
    print(res)
    if len(argv) > 1:
        open(argv[1], "w").write(res + "\n")


if __name__ == "__main__":
    report(sys.argv)
