"""ORACLE -- test infrastructure, not product code.

CPU restatement (NumPy / SciPy-LAPACK, float64) of the reference's per-leaf GP arithmetic:
kernels.jl, means.jl, gaussianprocess.jl and AdvancedCholeskey.jl.chol_continue!.  Every function
cites the reference lines it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (deepstructuredmixtures_amd/) never does.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, and Julia is not
available to run it.  The restatement is pinned instead by analytic known
answers (n=1, n=2 closed forms), 50-digit mpmath evaluations, finite differences and SciPy/LAPACK
cross-checks committed under tests/golden/ (generator: tests/golden/make_golden.py).
Third-party arithmetic restated here from its published behaviour, versions unpinned in
Project.toml: Distances.pairwise(SqEuclidean) (|a|^2 + |b|^2 - 2 a.b, clamped at 0),
LinearAlgebra/LAPACK potrf/trtrs, StatsFuns.logsumexp.
"""
import numpy as np
import scipy.linalg as sla

EPS = 1e-8  # src/DeepStructuredMixtures.jl:27


# ------------------------------------------------------------------------------- kernel functions

class IsoSE:
    """src/kernels.jl:59-106"""
    kind = 0

    def __init__(self, logl, logs):
        self.logl, self.logs = float(logl), float(logs)
        self.dl = 0.0
        self.ds = 0.0

    def lengthscale(self):
        return np.exp(self.logl)            # :73

    def variance(self):
        return np.exp(2.0 * self.logs)      # :68

    def std(self):
        return np.exp(self.logs)            # :69

    def nl(self):
        return 1


class ArdSE:
    """src/kernels.jl:109-170"""
    kind = 1

    def __init__(self, logl, logs):
        self.logl = np.array(logl, dtype=np.float64).reshape(-1)
        self.logs = float(logs)
        self.dl = np.zeros_like(self.logl)
        self.ds = 0.0

    def lengthscale(self):
        return np.exp(self.logl)

    def variance(self):
        return np.exp(2.0 * self.logs)

    def std(self):
        return np.exp(self.logs)

    def nl(self):
        return self.logl.size


class IsoLinear:
    """src/kernels.jl:174-205: variance is a dummy 1.0 (:181-183)."""
    kind = 2

    def __init__(self, logl):
        self.logl = float(logl)
        self.dl = 0.0

    def lengthscale(self):
        return np.exp(self.logl)

    def variance(self):
        return 1.0

    def std(self):
        return 1.0

    def nl(self):
        return 1


def make_kernel(kind, loghyp):
    """Kernel object from the log-scale vector [logl..., logs] (variance slot ignored for IsoLinear)."""
    loghyp = np.asarray(loghyp, dtype=np.float64)
    if kind == 0:
        return IsoSE(loghyp[0], loghyp[1])
    if kind == 1:
        return ArdSE(loghyp[:-1], loghyp[-1])
    return IsoLinear(loghyp[0])


def sqeuclidean_pairwise(x1, x2, exact=False):
    """Distances.pairwise(SqEuclidean(), x1, x2, dims=1) as called at src/kernels.jl:83.

    As written: r = |a|^2 + |b|^2 - 2 a.b (one GEMM), clamped at 0.  exact=True accumulates squared
    differences directly (what the HIP kernel does); the two agree to rounding."""
    x1 = np.asarray(x1, dtype=np.float64)
    x2 = np.asarray(x2, dtype=np.float64)
    if exact:
        P = np.zeros((x1.shape[0], x2.shape[0]))
        for d in range(x1.shape[1]):
            u = x1[:, d][:, None] - x2[:, d][None, :]
            P += u * u
        return P
    sa = np.sum(x1 * x1, axis=1)
    sb = np.sum(x2 * x2, axis=1)
    return np.maximum(sa[:, None] + sb[None, :] - 2.0 * (x1 @ x2.T), 0.0)


def getdistancematrix(k, x1, x2=None, exact=False):
    """src/kernels.jl:55 (one-argument form), :83 IsoSE, :137-144 ArdSE, :194 IsoLinear."""
    x2 = x1 if x2 is None else x2
    if k.kind == 0:
        return sqeuclidean_pairwise(x1, x2, exact)
    if k.kind == 1:
        P = np.zeros((x1.shape[0], x2.shape[0], k.nl()))
        for d in range(k.nl()):
            P[:, :, d] = sqeuclidean_pairwise(x1[:, d:d + 1], x2[:, d:d + 1], exact)
        return P
    return np.asarray(x1) @ np.asarray(x2).T


def kernelmatrix_from_P(k, P):
    """kernelmatrix!(kernel, K, P): src/kernels.jl:21-27 (Iso), :39-49 (Ard, additive via umap!)."""
    if k.kind == 0:
        l = k.lengthscale() ** 2
        return k.variance() * np.exp(-0.5 * (P / l))           # rbfkernel :78, lmul! :25
    if k.kind == 1:
        ls = k.lengthscale() ** 2
        K = np.zeros(P.shape[:2])
        for d in range(P.shape[2]):
            K += np.exp(-0.5 * (P[:, :, d] / ls[d]))            # umap! accumulates :31-37
        return k.variance() * K
    l = k.lengthscale() ** 2
    return 1.0 * (P / l)                                        # linearkernel :189


def kernelmatrix(k, x1, x2=None, exact=False):
    """src/kernels.jl:15-18"""
    return kernelmatrix_from_P(k, getdistancematrix(k, x1, x2, exact))


def prior_diag(k, x):
    """diag(kernelmatrix(k, x, x))"""
    if k.kind == 0:
        return np.full(x.shape[0], k.variance())
    if k.kind == 1:
        return np.full(x.shape[0], k.variance() * k.nl())
    return np.sum(x * x, axis=1) / k.lengthscale() ** 2


# ------------------------------------------------------------------------------- AdvancedCholesky

def chol_continue(A, ki):
    """AdvancedCholesky.chol_continue!(A, ki) (src/AdvancedCholeskey.jl:152-174), ki 1-based as in
    Julia: A[1:ki-1, 1:ki-1] already holds a lower factor; returns (lower-triangular A, info)."""
    A = np.tril(np.array(A, dtype=np.float64))                   # tril! :156
    p = ki - 1
    if p > 0 and p < A.shape[0]:
        L11 = A[:p, :p]
        # A21 /= L11'  (:161)  ->  solve X L11^T = A21
        A[p:, :p] = sla.solve_triangular(L11, A[p:, :p].T, lower=True, trans="N").T
        A[p:, p:] -= np.tril(A[p:, :p] @ A[p:, :p].T)            # syrk!('L','N',-1,...) :167
    info = 0
    if p < A.shape[0]:
        C, info = sla.lapack.dpotrf(A[p:, p:], lower=1, clean=1)  # potrf!('L', C) :171
        A[p:, p:] = C
    return A, int(info)


# ------------------------------------------------------------------------------- GaussianProcess

class GaussianProcess:
    """src/gaussianprocess.jl:14-80.  y is stored mean-subtracted (:72-74); P is kept (:35,57)."""

    def __init__(self, x, y, mean, kernel, logNoise, exact_dist=False):
        self.x = np.asarray(x, dtype=np.float64)
        self.N, self.D = self.x.shape
        self.mean = float(mean)                                   # ConstMean(m), src/means.jl:7-9
        self.y = np.asarray(y, dtype=np.float64) - self.mean      # apply_subtract! src/means.jl:11-14
        self.kernel = kernel
        self.logNoise = float(logNoise)
        self.dnoise = 0.0
        self.exact_dist = exact_dist
        self.P = getdistancematrix(kernel, self.x, None, exact_dist)
        self.factors = np.zeros((self.N, self.N))
        self.alpha = np.zeros(self.N)
        self.info = 0

    def getnoise(self):
        return np.exp(2.0 * self.logNoise)                        # :39

    def noisy_kernel(self):
        F = kernelmatrix_from_P(self.kernel, self.P).copy()       # :83, :91
        F[np.diag_indices(self.N)] += self.getnoise() + EPS       # :94-98
        return F

    def solve_alpha(self):
        L = np.tril(self.factors)
        z = sla.solve_triangular(L, self.y, lower=True)
        self.alpha = sla.solve_triangular(L, z, lower=True, trans="T")   # :105
        return self.alpha

    def update_cholesky(self):
        """src/gaussianprocess.jl:82-108; potrf info is ignored there, we keep it."""
        F = self.noisy_kernel()
        C, info = sla.lapack.dpotrf(F, lower=1, clean=1)          # :101
        self.factors = C
        self.info = int(info)
        self.solve_alpha()
        return self

    def L(self):
        return np.tril(self.factors)

    def mll(self):
        """src/gaussianprocess.jl:163"""
        logdet = 2.0 * np.sum(np.log(np.diag(self.factors)))
        return -(np.dot(self.y, self.alpha) + logdet + np.log(2.0 * np.pi) * self.N) / 2.0

    def prediction(self, xtest, full_cov=False):
        """src/gaussianprocess.jl:110-137.  full_cov=True is the reference as written (n_t x n_t
        Sigma); otherwise only its diagonal (all that src/common.jl:136,147 consume)."""
        xt = np.asarray(xtest, dtype=np.float64)
        Knt = kernelmatrix(self.kernel, self.x, xt, self.exact_dist)          # :133
        mu = self.mean + Knt.T @ self.alpha                                   # :117-118
        V = sla.solve_triangular(self.L(), Knt, lower=True)                   # :120
        if full_cov:
            Ktt = kernelmatrix(self.kernel, xt, xt, self.exact_dist)          # :134
            S = Ktt - V.T @ V                                                 # :121
            S[np.diag_indices(xt.shape[0])] += self.getnoise()                # :123-126
            return mu, S
        var = prior_diag(self.kernel, xt) - np.sum(V * V, axis=0) + self.getnoise()
        return mu, var

    def updategradients(self):
        """src/gaussianprocess.jl:165-178 + 219-226 and the kernel methods src/kernels.jl:85-99,
        146-164,196-200.  Returns the vector of src/gaussianprocess.jl:212-214: [dl..., ds, dnoise]."""
        K = kernelmatrix_from_P(self.kernel, self.P).copy()
        n = self.N
        L = self.L()
        Kinv = sla.cho_solve((L, True), np.eye(n))
        precomp = np.outer(self.alpha, self.alpha) - Kinv         # ααinvcK! :219-226
        self.dnoise = self.getnoise() * np.trace(precomp)         # :176
        k = self.kernel
        if k.kind == 0:
            s = k.std()
            l = k.lengthscale() ** 2
            K *= s                                                # lmul!(σ, K) :90
            k.ds = 0.5 * np.trace((precomp * 2.0) @ K)            # :93
            K *= self.P / l                                       # :96
            k.dl = 0.5 * np.trace(precomp @ K)                    # :97
            return np.array([k.dl, k.ds, self.dnoise])
        if k.kind == 1:
            s = k.std()
            ls = k.lengthscale() ** 2
            K *= s
            k.ds = 0.5 * np.trace((precomp * 2.0) @ K)            # :157
            PK = precomp @ K
            for d in range(k.nl()):
                # `precomp * K .* (p/ls[d])` parses as (precomp*K) .* (p/ls[d]); p has a zero diagonal,
                # so the trace vanishes identically (:161)
                k.dl[d] = 0.5 * np.trace(PK * (self.P[:, :, d] / ls[d]))
            return np.concatenate([k.dl, [k.ds, self.dnoise]])
        k.dl = 0.5 * np.trace((precomp * -2.0) @ K)               # :198
        return np.array([k.dl, 0.0, self.dnoise])                 # getgradients :201

    def grad(self):
        """∇mll(gp): src/gaussianprocess.jl:185-190"""
        return self.updategradients()
