"""Emulation of R's default random number generators (test infrastructure only).

The reference's tests, README and vignettes all create their inputs with
``set.seed(1234)`` followed by ``rnorm()`` / ``sample()``.  R is not installed in
the build container, so the golden inputs are regenerated here from R's published
algorithms (R sources ``src/main/RNG.c``, ``src/nmath/snorm.c``, ``src/main/random.c``,
R >= 3.6.0 defaults):

* ``set.seed(s)``: LCG scrambling ``seed = 69069 * seed + 1`` 50 times, then 625
  more draws fill ``i_seed``; ``i_seed[0]`` (the MT position) is forced to 624.
* ``unif_rand()``: MT19937 ``genrand_int32() * 2.3283064365386963e-10`` with the
  (0, 1) fix-up.
* ``norm_rand()``: "Inversion": ``u = (int)(2^27 * U1) + U2; qnorm(u / 2^27)``.
* ``sample(n, k)``: "Rejection" ``R_unif_index`` + the swap-with-last loop.

``qnorm`` is taken from ``scipy.special.ndtri`` (<= 1 ulp from R's AS241), which can
only matter if two draws were within 1 ulp of each other.

Only tests/ and tests/golden/make_golden.py import this module.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import ndtri

_I2_32M1 = 2.328306437080797e-10
_BIG = 134217728.0  # 2^27


class RRandom:
    """R's Mersenne-Twister + Inversion + Rejection generators after set.seed()."""

    def __init__(self, seed: int):
        self.set_seed(seed)

    def set_seed(self, seed: int) -> None:
        s = np.uint32(seed & 0xFFFFFFFF)
        mul = np.uint32(69069)
        one = np.uint32(1)
        with np.errstate(over="ignore"):
            for _ in range(50):
                s = s * mul + one
            state = np.empty(625, dtype=np.uint32)
            for j in range(625):
                s = s * mul + one
                state[j] = s
        self._bitgen = np.random.MT19937()
        self._bitgen.state = {
            "bit_generator": "MT19937",
            "state": {"key": state[1:].copy(), "pos": 624},
        }

    # -- uniform ---------------------------------------------------------------
    def unif_rand(self, size: int | None = None):
        if size is None:
            return float(self._fixup(self._bitgen.random_raw(1).astype(np.float64) * 2.3283064365386963e-10)[0])
        raw = self._bitgen.random_raw(size).astype(np.float64)
        return self._fixup(raw * 2.3283064365386963e-10)

    @staticmethod
    def _fixup(x: np.ndarray) -> np.ndarray:
        x = np.where(x <= 0.0, 0.5 * _I2_32M1, x)
        x = np.where(1.0 - x <= 0.0, 1.0 - 0.5 * _I2_32M1, x)
        return x

    # -- normal ----------------------------------------------------------------
    def rnorm(self, n: int, mean: float = 0.0, sd: float = 1.0) -> np.ndarray:
        u = self.unif_rand(2 * n)
        u1 = u[0::2]
        u2 = u[1::2]
        v = np.floor(_BIG * u1) + u2
        z = ndtri(v / _BIG)
        return mean + sd * z

    # -- sample ----------------------------------------------------------------
    def _rbits(self, bits: int) -> int:
        v = 0
        n = 0
        while n <= bits:
            v1 = int(math.floor(self.unif_rand() * 65536))
            v = 65536 * v + v1
            n += 16
        if bits < 64:
            v &= (1 << bits) - 1
        return v

    def unif_index(self, dn: int) -> int:
        if dn <= 0:
            return 0
        bits = int(math.ceil(math.log2(dn)))
        while True:
            dv = self._rbits(bits)
            if dv < dn:
                return dv

    def sample(self, n: int, k: int) -> np.ndarray:
        """R's sample(n, k) (without replacement), 1-based values."""
        x = list(range(n))
        out = np.empty(k, dtype=np.int64)
        nn = n
        for i in range(k):
            j = self.unif_index(nn)
            out[i] = x[j] + 1
            nn -= 1
            x[j] = x[nn]
        return out
