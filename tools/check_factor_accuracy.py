"""Backward error of the device Cholesky on single GPs: ||K_y - L L^T||_F / ||K_y||_F and max |L - L_lapack| / max |L|,
with K_y built from the device's own kernel matrix (so only the factorisation is measured).  Diagnostic."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepstructuredmixtures_amd import hipabi

ctx = hipabi.Context(0)
rng = np.random.default_rng(3)
for n, D, logl, lognoise in ((128, 2, np.log(0.3), np.log(0.1)), (500, 3, np.log(0.5), np.log(0.01)), (2000, 4, np.log(0.4), np.log(0.05)),
                            (3000, 2, np.log(0.8), np.log(0.003))):
    X = np.asfortranarray(rng.random((n, D)))
    y = rng.standard_normal(n)
    ctx.set_train(X, y)
    ctx.set_leaves(np.array([0, n]), np.arange(n), [0], [0.0])
    ctx.set_sharing(None, None, None)
    ctx.set_hyper(0, 0, [logl, 0.0, lognoise])
    mll, info, _ = ctx.fit()
    F, alpha = ctx.download_factor(0, n)
    L = np.tril(F)
    K = ctx.kernel_matrix(0, X, X) + (np.exp(2 * lognoise) + 1e-8) * np.eye(n)
    back = np.linalg.norm(K - L @ L.T) / np.linalg.norm(K)
    Lr = np.linalg.cholesky(K)
    print(f"n={n:5d} cond~{np.linalg.cond(K):.1e}  info={int(info[0])}  backward {back:.2e}   "
          f"max|L-L_lapack|/max|L| {np.max(np.abs(L - Lr)) / np.max(np.abs(Lr)):.2e}   lapack backward "
          f"{np.linalg.norm(K - Lr @ Lr.T) / np.linalg.norm(K):.2e}")
