"""Kernel- and mean-function parameter objects (host side).

Mirrors the parameter containers of the reference (`src/kernels.jl:59-76,109-131,174-187`,
`src/means.jl:7-18`). They hold hyper-parameters only: all kernel-matrix arithmetic runs in
the HIP library (`csrc/`), reached through the C ABI in `include/dsmgp_hip.h`.

Parametrisation (reference `src/kernels.jl:68-73`, `src/gaussianprocess.jl:39`):
  lengthscale = exp(logl), signal variance = exp(2*logs), noise variance = exp(2*logNoise).
Hyper-vector layout per kernel id: [logl..., logs, logNoise] (`src/gaussianprocess.jl:153-161`).
"""
import numpy as np

# numeric kinds shared with include/dsmgp_hip.h
KIND_ISO_SE = 0
KIND_ARD_SE = 1
KIND_ISO_LINEAR = 2


class KernelFunction:
    kind = -1

    def loghyp(self):
        """[logl..., logs] on the log scale (variance slot is a dummy 0.0 for IsoLinear)."""
        raise NotImplementedError

    def set_loghyp(self, v):
        raise NotImplementedError

    def nparams(self):
        return len(self.loghyp())

    def copy(self):
        raise NotImplementedError


class IsoSE(KernelFunction):
    """k(a,b) = exp(2 logs) * exp(-0.5 |a-b|^2 / exp(logl)^2)  (`src/kernels.jl:59-83`)."""
    kind = KIND_ISO_SE

    def __init__(self, logl, logs):
        self.logl = float(logl)
        self.logs = float(logs)
        self.dl = 0.0
        self.ds = 0.0

    def loghyp(self):
        return np.array([self.logl, self.logs])

    def set_loghyp(self, v):
        self.logl, self.logs = float(v[0]), float(v[1])

    def copy(self):
        return IsoSE(self.logl, self.logs)

    def __repr__(self):
        return f"IsoSE({self.logl}, {self.logs})"


class ArdSE(KernelFunction):
    """ADDITIVE ARD kernel exp(2 logs) * sum_d exp(-0.5 (a_d-b_d)^2 / exp(logl_d)^2)
    (`src/kernels.jl:31-49,109-144`: `umap!` accumulates per dimension)."""
    kind = KIND_ARD_SE

    def __init__(self, logl, logs):
        self.logl = np.array(logl, dtype=np.float64).reshape(-1)
        self.logs = float(logs)
        self.dl = np.zeros_like(self.logl)
        self.ds = 0.0

    def loghyp(self):
        return np.concatenate([self.logl, [self.logs]])

    def set_loghyp(self, v):
        self.logl = np.array(v[:-1], dtype=np.float64)
        self.logs = float(v[-1])

    def copy(self):
        return ArdSE(self.logl.copy(), self.logs)

    def __repr__(self):
        return f"ArdSE({self.logl.tolist()}, {self.logs})"


class IsoLinear(KernelFunction):
    """k(a,b) = a.b / exp(logl)^2; the variance slot is a dummy (`src/kernels.jl:174-194`)."""
    kind = KIND_ISO_LINEAR

    def __init__(self, logl):
        self.logl = float(logl)
        self.dl = 0.0

    def loghyp(self):
        return np.array([self.logl, 0.0])

    def set_loghyp(self, v):
        self.logl = float(v[0])  # setvariance! is a no-op (`src/kernels.jl:183`)

    def copy(self):
        return IsoLinear(self.logl)

    def __repr__(self):
        return f"IsoLinear({self.logl})"


class ConstMean:
    """Constant mean function (`src/means.jl:7-18`)."""

    def __init__(self, m):
        self.m = float(m)

    def __repr__(self):
        return f"ConstMean({self.m})"
