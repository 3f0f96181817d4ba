"""Shared definition of the golden cases: seeded inputs, the oracle run, and the digest format."""
import hashlib

import numpy as np

from oracle import oracle as orc

CASES = {
    "ntt_n10": {"op": "ntt", "logN": 10, "L": 3, "K": 2, "idx": [0, 1, 2, 3, 4], "seed": 101},
    "ntt_n13": {"op": "ntt", "logN": 13, "L": 4, "K": 4, "idx": [0, 3, 4, 7], "seed": 102},
    "modup_n10": {"op": "mod_up", "logN": 10, "L": 4, "K": 4, "Li": 4, "seed": 103},
    "moddown_n10": {"op": "mod_down", "logN": 10, "L": 4, "K": 4, "Li": 3, "seed": 104},
    "rescale_n10": {"op": "rescale", "logN": 10, "L": 4, "K": 1, "Li": 4, "seed": 105},
    "galois_n10": {"op": "galois", "logN": 10, "L": 3, "K": 1, "Li": 3, "galois": [3, 9, 1535], "seed": 106},
    "mulrelin_n10": {"op": "mul_relin_rescale", "logN": 10, "L": 4, "K": 1, "Li": 4, "seed": 107},
    "mulrelin_n12_li3": {"op": "mul_relin_rescale", "logN": 12, "L": 4, "K": 1, "Li": 3, "seed": 108},
    "rotate_n10": {"op": "rotate", "logN": 10, "L": 4, "K": 4, "Li": 4, "offset": 1, "seed": 109},
    "rotate_n11_li2": {"op": "rotate", "logN": 11, "L": 3, "K": 3, "Li": 2, "offset": 4, "seed": 110},
}


def rand_poly(o, idx, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, int(o.primes[i]), o.N, dtype=np.uint64) for i in idx])


def case_inputs(case):
    o = orc.Oracle(case["logN"], case["L"], case["K"])
    op, seed, inp = case["op"], case["seed"], {}
    Li = case.get("Li", 0)
    if op == "ntt":
        inp["x"] = rand_poly(o, case["idx"], seed)
    elif op in ("mod_up", "rescale", "galois"):
        inp["x"] = rand_poly(o, range(Li), seed)
    elif op == "mod_down":
        inp["x"] = rand_poly(o, list(range(Li)) + list(range(case["L"], case["L"] + case["K"])), seed)
    elif op == "mul_relin_rescale":
        sx, _ = o.keygen_sk(seed=seed)
        inp["beta"], inp["alpha"] = o.keygen_mult(sx, seed=seed + 1)
        inp["c0"] = np.stack([rand_poly(o, range(Li), seed + 2), rand_poly(o, range(Li), seed + 3)])
        inp["c1"] = np.stack([rand_poly(o, range(Li), seed + 4), rand_poly(o, range(Li), seed + 5)])
    elif op == "rotate":
        sx, ext = o.keygen_sk(seed=seed)
        inp["galois"] = o.rotation_galois(case["offset"])
        inp["rkb"], inp["rka"] = o.keygen_rot(sx, ext, inp["galois"], seed=seed + 1)
        inp["ct"] = np.stack([rand_poly(o, range(Li), seed + 2), rand_poly(o, range(Li), seed + 3)])
    return o, inp


def run_case_oracle(case):
    o, inp = case_inputs(case)
    op = case["op"]
    if op == "ntt":
        f = o.ntt_forward(inp["x"].copy(), case["idx"])
        return {"forward": f, "backward": o.ntt_backward(inp["x"].copy(), case["idx"])}
    if op == "mod_up":
        return {"out": o.mod_up(inp["x"])}
    if op == "mod_down":
        return {"out": o.mod_down(inp["x"])}
    if op == "rescale":
        return {"out": o.rescale(inp["x"])}
    if op == "galois":
        res = {}
        for g in case["galois"]:
            res["ntt_%d" % g] = o.apply_galois(inp["x"].copy(), g)
            res["coeff_%d" % g] = o.apply_galois_non_ntt(inp["x"].copy(), g)
        return res
    if op == "mul_relin_rescale":
        ob, oa = o.mul_relin_rescale(inp["c0"][0], inp["c0"][1], inp["c1"][0], inp["c1"][1], inp["beta"], inp["alpha"])
        return {"bx": ob, "ax": oa}
    if op == "rotate":
        rb, ra = o.rotate(inp["ct"][0], inp["ct"][1], inp["galois"], inp["rkb"], inp["rka"])
        return {"bx": rb, "ax": ra}
    raise ValueError(op)


def run_case_hip(case):
    import torch
    import fhe_amd
    o, inp = case_inputs(case)
    g = fhe_amd.HipContext(case["logN"], case["L"], case["K"], o.primes, o.psi)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()  # noqa: E731
    host = lambda t: t.cpu().numpy().view(np.uint64)  # noqa: E731
    op, Li, N, K = case["op"], case.get("Li", 0), o.N, case["K"]
    try:
        if op == "ntt":
            f, b = dev(inp["x"]), dev(inp["x"])
            g.ntt_forward(f, case["idx"])
            g.ntt_backward(b, case["idx"])
            return {"forward": host(f), "backward": host(b)}
        if op == "mod_up":
            r = torch.zeros((Li + K, N), dtype=torch.int64, device="cuda")
            return {"out": host(g.mod_up(r, dev(inp["x"]), Li))}
        if op == "mod_down":
            r = torch.zeros((Li, N), dtype=torch.int64, device="cuda")
            return {"out": host(g.mod_down(r, dev(inp["x"]), Li))}
        if op == "rescale":
            r = torch.zeros((Li - 1, N), dtype=torch.int64, device="cuda")
            return {"out": host(g.rescale(r, dev(inp["x"]), Li))}
        if op == "galois":
            res = {}
            for gal in case["galois"]:
                res["ntt_%d" % gal] = host(g.apply_galois(dev(inp["x"]), Li, gal))
                res["coeff_%d" % gal] = host(g.apply_galois_non_ntt(dev(inp["x"]), Li, gal))
            return res
        if op == "mul_relin_rescale":
            g.load_mult_key(dev(inp["beta"]), dev(inp["alpha"]))
            out = torch.zeros((1, 2, Li - 1, N), dtype=torch.int64, device="cuda")
            g.mul_relin_rescale(out, dev(inp["c0"][None]), dev(inp["c1"][None]), Li)
            return {"bx": host(out)[0, 0], "ax": host(out)[0, 1]}
        if op == "rotate":
            g.load_rot_key(inp["galois"], dev(inp["rkb"]), dev(inp["rka"]))
            ct = dev(inp["ct"][None])
            g.rotate(inp["galois"], ct, Li)
            return {"bx": host(ct)[0, 0], "ax": host(ct)[0, 1]}
    finally:
        g.close()
    raise ValueError(op)


def digest_outputs(outs):
    res = {}
    for name, arr in sorted(outs.items()):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        flat = arr.reshape(arr.shape[0], -1)
        rng = np.random.default_rng(len(name) + flat.size)
        samples = []
        for _ in range(16):
            li, d = int(rng.integers(0, flat.shape[0])), int(rng.integers(0, flat.shape[1]))
            samples.append([li, d, str(int(flat[li, d]))])
        res[name] = {"shape": list(arr.shape), "sha256": hashlib.sha256(arr.astype("<u8").tobytes()).hexdigest(), "samples": samples}
    return res
