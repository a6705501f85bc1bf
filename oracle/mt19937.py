"""ORACLE (test infrastructure only — never imported by the product path).

MT19937 and the two draw disciplines the reference's sampler uses
(/root/reference/utils/replay_buffer.py:168-170 `random.choice`, :222 `np.random.randint`).
The algorithms live in third-party code that is not vendored in the reference:
  * CPython 3.10 `random` (Lib/random.py `_randbelow_with_getrandbits`, Modules/_randommodule.c
    `init_by_array`, `getrandbits(k) = genrand_uint32() >> (32-k)` for k <= 32);
  * NumPy legacy `RandomState` (`_legacy_seeding` -> `init_genrand`; `randint` int64 path ->
    masked rejection `while ((v = next_uint32() & mask) > rng)`, no draw when rng == 0).
Pinned by tests/golden/replay_*.npz (picks/starts recorded from the live generators).
"""
import numpy as np

N, M = 624, 397
_U = np.uint32


class MT19937:
    def __init__(self):
        self.mt = np.zeros(N, dtype=np.uint32)
        self.idx = N

    # ---- seeding -------------------------------------------------------------------------
    def init_genrand(self, s):
        mt = [0] * N
        mt[0] = s & 0xFFFFFFFF
        for i in range(1, N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt = np.array(mt, dtype=np.uint32)
        self.idx = N
        return self

    def init_by_array(self, key):
        self.init_genrand(19650218)
        mt = [int(x) for x in self.mt]
        i, j = 1, 0
        klen = len(key)
        for _ in range(max(N, klen)):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525)) + key[j] + j) & 0xFFFFFFFF
            i += 1
            j += 1
            if i >= N:
                mt[0] = mt[N - 1]
                i = 1
            if j >= klen:
                j = 0
        for _ in range(N - 1):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941)) - i) & 0xFFFFFFFF
            i += 1
            if i >= N:
                mt[0] = mt[N - 1]
                i = 1
        mt[0] = 0x80000000
        self.mt = np.array(mt, dtype=np.uint32)
        self.idx = N
        return self

    @classmethod
    def python_seed(cls, a):
        """random.seed(int): init_by_array over the 32-bit limbs of abs(a)."""
        a = abs(int(a))
        key = []
        while a:
            key.append(a & 0xFFFFFFFF)
            a >>= 32
        return cls().init_by_array(key or [0])

    @classmethod
    def numpy_seed(cls, s):
        """np.random.seed(int): init_genrand."""
        return cls().init_genrand(int(s))

    @classmethod
    def from_state(cls, key, pos):
        g = cls()
        g.mt = np.array(key, dtype=np.uint32).copy()
        g.idx = int(pos)
        return g

    # ---- generation ----------------------------------------------------------------------
    def _twist(self):
        mt = [int(x) for x in self.mt]
        for k in range(N):
            y = (mt[k] & 0x80000000) | (mt[(k + 1) % N] & 0x7FFFFFFF)
            v = mt[(k + M) % N] ^ (y >> 1)
            if y & 1:
                v ^= 0x9908B0DF
            mt[k] = v
        self.mt = np.array(mt, dtype=np.uint32)
        self.idx = 0

    def genrand_uint32(self):
        if self.idx >= N:
            self._twist()
        y = int(self.mt[self.idx])
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    # ---- draw disciplines ----------------------------------------------------------------
    def py_randbelow(self, n):
        """CPython random._randbelow_with_getrandbits(n), n < 2**32."""
        if n == 0:
            return 0
        k = int(n).bit_length()
        r = self.genrand_uint32() >> (32 - k)
        while r >= n:
            r = self.genrand_uint32() >> (32 - k)
        return r

    def np_randint0(self, hi):
        """NumPy legacy RandomState.randint(0, hi) (int64 path, rng < 2**32)."""
        rng = int(hi) - 1
        if rng == 0:
            return 0
        mask = rng
        for s in (1, 2, 4, 8, 16):
            mask |= mask >> s
        v = self.genrand_uint32() & mask
        while v > rng:
            v = self.genrand_uint32() & mask
        return v
