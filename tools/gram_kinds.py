import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepstructuredmixtures_amd as dsm
X, y, Xt = dsm.regression_data(100000, 8, seed=20204)
for name, kern in (("IsoSE", dsm.IsoSE(np.log(0.3), 0.0)), ("IsoLinear", dsm.IsoLinear(np.log(1.5))), ("ArdSE", dsm.ArdSE(np.log(np.full(8, 0.3)), 0.0))):
    m = dsm.buildDSMGP(X, y, 3, 4, M=200, D=2, kernel=kern, logNoise=np.log(0.1), seed=20204, fit_now=False)
    m.ctx.set_profile(2)
    try:
        dsm.fit(m); dsm.fit(m)
    except Exception as e:
        print(name, "fit failed:", str(e)[:60])
    t = m.ctx.timings()
    print(name, "gram %.4f s  total_fit %.4f" % (t["gram"], t["total_fit"]), flush=True)
    m.ctx.close()
