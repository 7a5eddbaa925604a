"""predict(model, x) on rows the model has not seen, phase by phase (routing on the host, registration, sweep + aggregation):
    python tools/time_predict_new_rows.py [config]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import deepstructuredmixtures_amd as dsm
from deepstructuredmixtures_amd import tree as ptree

cfg = sys.argv[1] if len(sys.argv) > 1 else "dsmgp_n100k_d8_depth4"
model, X, y, Xt, ptr, idx = bench.build_model(cfg, 0, 1, 0)
dsm.fit(model)
dsm.update(model)
dsm.predict(model, Xt)                                     # first use: arenas, Dinv completion
for rep in range(6):
    x = np.ascontiguousarray(Xt[::-1] if rep % 2 == 0 else Xt)
    t0 = time.perf_counter(); ptree.route_recursive(model.root, x); t1 = time.perf_counter()
    ptree.route(model.root, x); t2 = time.perf_counter()
    mu, var = dsm.predict(model, x); t3 = time.perf_counter()
    print(f"{cfg}: routing by recursion {t1 - t0:.4f} s, by the library's host routine {t2 - t1:.4f} s; predict on new rows {t3 - t2:.4f} s")
