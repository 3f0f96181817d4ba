"""TEST INFRASTRUCTURE ONLY — numpy/Python restatement of the reference's CKKS encoder (src/encoder.cpp:5-200) used to check
the host mirror's fHE::Encoder.  Floating point: compared within a tolerance stated in the test.
PARITY UNPINNED: the reference holds no fixture for the encoder (and its prime table lives in the absent yell library)."""
import numpy as np


def _brv(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


class EncoderOracle:
    def __init__(self, logN):
        self.logn = logN
        self.N = 1 << logN
        self.slots = self.N >> 1
        m = self.N << 1
        idx = np.zeros(self.N, np.int64)
        pos = 1
        for i in range(self.slots):  # encoder.cpp:9-25
            idx[i] = _brv((pos - 1) >> 1, logN)
            idx[self.slots | i] = _brv((m - pos - 1) >> 1, logN)
            pos = (pos * 3) & (m - 1)
        self.index_map = idx
        k = np.array([_brv(i, logN) for i in range(self.N)], dtype=np.float64)
        self.roots = np.exp(2j * np.pi * k / m)  # psi^brv(i), encoder.cpp:27-33
        self.inv_roots = 1.0 / self.roots

    def _ifft(self, res):  # encoder.cpp:60-83
        tt = 1
        for i in range(self.logn):
            h = (1 << (self.logn - i)) >> 1
            v = res.reshape(h, 2, tt)
            s = self.inv_roots[h:2 * h].reshape(h, 1)
            u, w = v[:, 0, :].copy(), v[:, 1, :].copy()
            v[:, 0, :] = u + w
            v[:, 1, :] = (u - w) * s
            tt *= 2
        return res

    def _fft(self, res):  # encoder.cpp:38-58
        tt = self.N
        for i in range(self.logn):
            mm = 1 << i
            tt >>= 1
            v = res.reshape(mm, 2, tt)
            s = self.roots[mm:2 * mm].reshape(mm, 1)
            u, w = v[:, 0, :].copy(), v[:, 1, :] * s
            v[:, 0, :] = u + w
            v[:, 1, :] = u - w
        return res

    def encode(self, values, scale, primes):
        """values: real or complex, <= N/2 entries -> [len(primes)][N] uint64, coefficient = trunc(x) mod p (encoder.cpp:116-147)."""
        vals = np.asarray(values, dtype=np.complex128)
        cv = np.zeros(self.N, np.complex128)
        n = len(vals)
        cv[self.index_map[:n]] = vals
        cv[self.index_map[self.slots:self.slots + n]] = np.conj(vals)
        cv = self._ifft(cv)
        scaled = cv.real * (scale / self.N)
        ints = [int(x) for x in scaled]  # truncation toward zero, like mpz_set_d
        return np.array([[c % int(p) for c in ints] for p in primes], dtype=np.uint64)

    def decode(self, limbs, primes, scale):
        """[n][N] uint64 -> N/2 complex slots (encoder.cpp:159-188)."""
        primes = [int(p) for p in primes]
        Q = 1
        for p in primes:
            Q *= p
        # CRT constants once per call (same integers as recomputing them per coefficient)
        Qk = [Q // p for p in primes]
        term = [Qk[k] * pow(Qk[k], -1, p) for k, p in enumerate(primes)]
        cols = [[int(v) for v in limbs[k]] for k in range(len(primes))]
        coef = []
        for i in range(self.N):
            x = 0
            for k in range(len(primes)):
                x += cols[k][i] * term[k]
            x %= Q
            if x >= Q // 2:
                x -= Q
            coef.append(float(x) / scale)
        res = self._fft(np.array(coef, dtype=np.complex128))
        return res[self.index_map[:self.slots]]
