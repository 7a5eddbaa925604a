"""Where `model_build_s` of the bench line goes (host only, no device): one buildDSMGP in a FRESH process -- the bench
builds its model once, so first-call costs (page faults of the exported tables) count -- split into the native recursion
(dsmgp_tree_build), the export of its tables, the region means, the node objects made from the table, and the rest
(overlap, model).  `python tools/time_model_build.py [config] [repeats]`: one JSON line per build."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                      # noqa: E402
import bench                                            # noqa: E402
import deepstructuredmixtures_amd as dsm                # noqa: E402
from deepstructuredmixtures_amd import hipabi, tree as ptree   # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "dsmgp_n100k_d8_depth4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c, h = bench.CONFIGS[cfg], bench.HYPER["survey"]
X, y, Xt = dsm.regression_data(c["N"], c["D"], seed=20204)
kern = dsm.IsoSE(h["logl"], h["logs"])
acc = {}
lib = hipabi.load_library()


def wrap_lib(name):
    f = getattr(lib, name)

    def g(*a):
        t0 = time.perf_counter()
        r = f(*a)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(lib, name, g)


def wrap(mod, name, key):
    f = getattr(mod, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(mod, name, g)


for n in ("dsmgp_tree_build", "dsmgp_tree_export", "dsmgp_tree_means"):
    wrap_lib(n)
wrap(hipabi, "tree_build", "table")
wrap(ptree, "_build_tree_native", "tree")
for i in range(reps):
    acc.clear()
    t0 = time.perf_counter()
    m = dsm.buildDSMGP(X, y, c["K"], c["V"], M=c["M"], D=c["depth"], kernel=kern, logNoise=h["lognoise"], seed=20204,
                       fit_now=False, device=None)
    tot = time.perf_counter() - t0
    nat = acc["dsmgp_tree_build"]
    print(json.dumps({"config": cfg, "call": i, "model_build_s": round(tot, 4), "native_recursion": round(nat, 4),
                      "export": round(acc["dsmgp_tree_export"], 4), "means": round(acc.get("dsmgp_tree_means", 0.0), 4),
                      "table_other": round(acc["table"] - nat - acc["dsmgp_tree_export"] - acc.get("dsmgp_tree_means", 0.0), 4),
                      "node_objects": round(acc["tree"] - acc["table"], 4), "rest": round(tot - acc["tree"], 4),
                      "leaves": len(m.leaves)}), flush=True)
    del m
