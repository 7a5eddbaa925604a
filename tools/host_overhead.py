"""Wall-clock vs device time of one fit + update + predict step (where does the host spend time?)."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import deepstructuredmixtures_amd as dsm

cfg = sys.argv[1] if len(sys.argv) > 1 else "dsmgp_n100k_d8"
model, X, y, Xt, ptr, idx = bench.build_model(cfg, 0, 1, 0)
if "--resident" in sys.argv:
    dsm.resident_test(model, Xt)          # the bench's timed step: the test rows ride through the fit
for _ in range(2):
    dsm.fit(model); dsm.update(model); dsm.predict(model, Xt)
t0 = time.perf_counter(); dsm.fit(model); t1 = time.perf_counter(); dsm.update(model); t2 = time.perf_counter()
mu, var = dsm.predict(model, Xt); t3 = time.perf_counter()
print(f"fit wall {t1-t0:.4f} (device {model.last_fit_seconds:.4f})  update {t2-t1:.4f}  predict wall {t3-t2:.4f} (device {model.last_predict_seconds:.4f})")
pr = cProfile.Profile(); pr.enable()
dsm.fit(model); dsm.update(model); dsm.predict(model, Xt)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:6000])
