"""Portable counter-based synthetic data generator.

SplitMix64 in counter mode: draw i of stream `seed` is mix(seed + (i+1)*GAMMA), so
Python, C++ and Julia produce identical bits without sharing generator state.
Uniforms use the top 53 bits; normals are Box-Muller pairs on consecutive draws.

The reference has no generator of its own for regression inputs (README.md:35-38
uses Julia's `randn`, whose stream cannot be reproduced here); the shapes follow
BASELINE.md "Inputs".
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, start, count):
    """Draws `start .. start+count-1` (0-based) of stream `seed` as uint64."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed, start, count):
    """53-bit uniforms in [0, 1)."""
    return (splitmix64(seed, start, count) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed, start, count):
    """Box-Muller normals; draw pair j uses uniforms 2j, 2j+1 of the stream (offset `start`)."""
    m = (count + 1) // 2
    u = uniform(seed, start, 2 * m)
    u1 = 1.0 - u[0::2]  # (0, 1]
    u2 = u[1::2]
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.empty(2 * m)
    z[0::2] = r * np.cos(2.0 * np.pi * u2)
    z[1::2] = r * np.sin(2.0 * np.pi * u2)
    return z[:count]


class Stream:
    """Sequential view over one counter stream (used by the tree builder)."""

    def __init__(self, seed):
        self.seed = int(seed)
        self.pos = 0

    def rand(self, count=None):
        n = 1 if count is None else int(count)
        u = uniform(self.seed, self.pos, n)
        self.pos += n
        return float(u[0]) if count is None else u

    def randint(self, lo, hi):
        """Integer in [lo, hi] inclusive."""
        return lo + int(self.rand() * (hi - lo + 1))

    def beta22(self):
        """Beta(2,2) = median of three uniforms."""
        return float(np.sort(self.rand(3))[1])

    def categorical(self, p):
        """1 draw from Categorical(p), 0-based."""
        c = np.cumsum(np.asarray(p, dtype=np.float64))
        return int(min(np.searchsorted(c, self.rand() * c[-1], side="right"), len(c) - 1))

    def dirichlet1(self, k):
        """Dirichlet(1,...,1) of dimension k."""
        e = -np.log(1.0 - self.rand(k))
        return e / e.sum()


def regression_data(N, D, n_test=None, seed=20200):
    """X ~ U(0,1)^{N x D} filled column-major, y = mean_d sin(2*pi*(d+1)*x_d) + 0.1*z.

    Returns (X, y, Xt): X, Xt Fortran-ordered float64, Xt drawn from the next stream positions.
    """
    n_test = N // 10 if n_test is None else n_test
    X = np.asfortranarray(uniform(seed, 0, N * D).reshape((N, D), order="F"))
    f = np.zeros(N)
    for d in range(D):
        f += np.sin(2.0 * np.pi * (d + 1) * X[:, d])
    y = f / D + 0.1 * normal(seed + 1, 0, N)
    Xt = np.asfortranarray(uniform(seed, N * D, n_test * D).reshape((n_test, D), order="F"))
    return X, y, Xt
