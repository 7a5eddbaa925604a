"""Host time of update! + predict on a resident test set (no fit in between): where the per-step milliseconds outside the device go.
   python tools/host_predict_profile.py [config]"""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import deepstructuredmixtures_amd as dsm

cfg = sys.argv[1] if len(sys.argv) > 1 else "dsmgp_n100k_d8_depth4"
model, X, y, Xt, ptr, idx = bench.build_model(cfg, 0, 1, 0)
dsm.resident_test(model, Xt)
dsm.fit(model); dsm.update(model); dsm.predict(model, Xt)
for name, f in (("update", lambda: dsm.update(model)), ("predict", lambda: dsm.predict(model, Xt))):
    t0 = time.perf_counter()
    for _ in range(50):
        f()
    print(f"{name}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per call")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50):
        f()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:2600])
