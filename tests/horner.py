"""Sampled-point checks of the NTT definition at full ring sizes, shared by the CPU (oracle) and GPU (HIP) tests: the forward
NTT of x under prime p is  X[d] = x(psi^(2*brv(d)+1)) mod p  (reference src/yell.cpp:16-21 fixes this ordering), and one
output needs only N modular multiplications by Horner's rule in Python integers, independent of both implementations."""


def brv(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2)


def horner_eval(coeffs, point, p):
    """coeffs[0] + coeffs[1]*point + ... mod p, Python integers."""
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * point + c) % p
    return acc


def ntt_value_at(coeffs, d, logN, psi, p):
    """The forward-NTT output at (bit-reversed) position d of the polynomial with the given natural-order coefficients."""
    return horner_eval(coeffs, pow(psi, 2 * brv(d, logN) + 1, p), p)


def sample_positions(logN, rng, count=8):
    N = 1 << logN
    fixed = [0, 1, N - 1, N // 2]  # first, second, last, and the first element of the second half
    rest = [int(v) for v in rng.integers(0, N, max(0, count - len(fixed)))]
    return fixed + rest
