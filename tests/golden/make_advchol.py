"""Golden fixture built the way the reference's own in-source self-checks build their inputs
(src/AdvancedCholeskey.jl: genCov :12, lrtest :61-110, test_chol_continue :121-135) -- the only executable checks the
reference holds for this path.  Julia's `rand` stream cannot be reproduced here, so the uniform entries come from the
repo's portable counter-based generator (datagen.uniform); everything else follows the reference's construction:

    genCov(D)            Sigma = Symmetric(rand(D, D) .+ D*I, :L)          lower triangle of the random matrix is used
    test_chol_continue   potrf!('L', Sigma[1:P, 1:P]); chol_continue!(Sigma, P+1)  ==  cholesky(Sigma)
    lrtest               B = A[idx, idx] with 10 rows removed; the factor of B the row-deletion path must reproduce
                         is cholesky(B)  (that path is defective in the reference, SURVEY F4: we factorise B in full)

Expected factors come from LAPACK dpotrf (SciPy) -- the same third-party routine the reference calls
(`LinearAlgebra.cholesky!` -> LAPACK.potrf!) -- and, for the D = 100 case, are checked against a 50-digit mpmath
Cholesky before they are written.  Run from the repo root:  python tests/golden/make_advchol.py
"""
import os
import sys

import mpmath as mp
import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepstructuredmixtures_amd.datagen import uniform, splitmix64  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def gen_cov(D, seed):
    """genCov(D; uplo=:L), src/AdvancedCholeskey.jl:12 (column-major fill like Julia's rand(D, D))."""
    R = uniform(seed, 0, D * D).reshape((D, D), order="F") + D * np.eye(D)
    return np.tril(R) + np.tril(R, -1).T


def missing_rows(D, seed, k=10):
    """shuffle(1:(D-1))[1:10] (src/AdvancedCholeskey.jl:63), 0-based here: k distinct rows of 0..D-2."""
    keys = splitmix64(seed, 0, D - 1)
    return np.sort(np.argsort(keys, kind="stable")[:k])


def main():
    out = {}
    # ---- test_chol_continue at the reference's own size (D = 100, P = 10) and at sizes that cross the 128-block edge
    for name, D, P, seed in (("cont_d100_p10", 100, 10, 9001), ("cont_d192_p150", 192, 150, 9002)):
        S = gen_cov(D, seed)
        L = np.linalg.cholesky(S)
        C, info = sla.lapack.dpotrf(S, lower=1, clean=1)
        assert info == 0 and np.max(np.abs(C - L)) < 1e-13
        if D == 100:
            mp.mp.dps = 50
            Lmp = mp.cholesky(mp.matrix(S.tolist()))
            Lm = np.array([[float(Lmp[i, j]) for j in range(D)] for i in range(D)])
            assert np.max(np.abs(Lm - C)) < 1e-13, np.max(np.abs(Lm - C))
            C = Lm
        out[f"{name}/D"], out[f"{name}/P"], out[f"{name}/seed"] = D, P, seed
        out[f"{name}/L"] = C
    # ---- lrtest's construction at its own size D = 1000: ten rows removed; summary of cholesky(B) (a full factor
    #      would be 8 MB): diagonal, log-determinant, Frobenius norm and three sampled columns
    D, seed = 1000, 9003
    A = gen_cov(D, seed)
    miss = missing_rows(D, seed + 1)
    idx = np.setdiff1d(np.arange(D), miss)
    for tag, M in (("lr_A", A), ("lr_B", A[np.ix_(idx, idx)])):
        C, info = sla.lapack.dpotrf(M, lower=1, clean=1)
        assert info == 0
        out[f"{tag}/diag"] = np.diag(C).copy()
        out[f"{tag}/logdet"] = 2.0 * np.sum(np.log(np.diag(C)))
        out[f"{tag}/fro"] = np.linalg.norm(C)
        out[f"{tag}/cols"] = np.array([0, 517, M.shape[0] - 3])
        out[f"{tag}/colvals"] = C[:, [0, 517, M.shape[0] - 3]].copy()
    out["lr/D"], out["lr/seed"], out["lr/missing"] = D, seed, miss
    np.savez_compressed(os.path.join(OUT, "advchol.npz"), **out)
    print("wrote advchol.npz:", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
