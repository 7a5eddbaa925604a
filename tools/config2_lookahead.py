"""Config 2 (single exact GP, n = 4096: 32 dependent block steps) with and without the lookahead schedule (diagnostic)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepstructuredmixtures_amd as dsm
from deepstructuredmixtures_amd import hipabi

X, y, Xt = dsm.regression_data(4096, 4, seed=20202)
for look in (0, 1, 0, 1):
    gp = dsm.GaussianProcess(X, y, kernel=dsm.IsoSE(np.log(0.3), 0.0), logNoise=np.log(0.1))
    gp.model.ctx.set_option(hipabi.OPT_LOOKAHEAD, look)
    def step():
        dsm.update_cholesky(gp); return dsm.prediction(gp, Xt)
    step(); step()
    t0 = time.perf_counter()
    for _ in range(10):
        mu, var = step()
    print(f"lookahead {look}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per update_cholesky! + prediction   mu[0] {mu[0]:.12f}")
