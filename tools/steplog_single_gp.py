import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepstructuredmixtures_amd as dsm
X, y, Xt = dsm.regression_data(4096, 4, seed=20202)
gp = dsm.GaussianProcess(X, y, kernel=dsm.IsoSE(np.log(0.3), 0.0), logNoise=np.log(0.1))
gp.model.ctx.set_profile(2)
dsm.update_cholesky(gp); dsm.update_cholesky(gp)
print({k: round(v * 1e3, 3) for k, v in gp.model.ctx.timings().items() if v > 0})
