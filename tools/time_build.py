"""Host-side set-up time of a bench config, phase by phase (no device): tree, overlap, sharing schedule, routing.
   python tools/time_build.py [config] [repeats]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402
import deepstructuredmixtures_amd as dsm  # noqa: E402
from deepstructuredmixtures_amd import tree  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "dsmgp_n100k_d8_depth4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = bench.CONFIGS[cfg]
for i in range(3):      # what a first touch of fresh memory costs on this machine
    t0 = time.perf_counter()
    a = np.empty(8_100_000, dtype=np.int64)
    a[:] = 1
    print(f"# first touch of 65 MB: {time.perf_counter() - t0:.3f} s")
    keep = a if i == 0 else None
    del a
X, y, Xt = dsm.regression_data(c["N"], c["D"], seed=20204)
kern = dsm.IsoSE(np.log(0.3), 0.0)
if c.get("kvec"):
    kern = [kern, dsm.IsoLinear(np.log(1.5))]
config = tree.DSMGPConfig(None, kern, np.log(0.1), c["M"], c["V"], c["K"], c["depth"], 0.5, True)
for rep in range(reps):
    t = [time.perf_counter()]
    root = tree.build_tree(X, y, config, seed=20204)
    t.append(time.perf_counter())
    leaves = tree.get_leaves(root)
    ov = tree.get_overlap(root, len(leaves))
    t.append(time.perf_counter())
    op, src, _ = tree.share_schedule(leaves, ov, 0.05)
    t.append(time.perf_counter())
    ptr, idx = tree.route(root, Xt)
    t.append(time.perf_counter())
    d = np.diff(t)
    print(f"{cfg}: L={len(leaves)} tree {d[0]:.3f} s, leaves+overlap {d[1]:.3f} s, schedule {d[2]:.3f} s, route {d[3]:.3f} s"
          f"  (build = {d[0] + d[1]:.3f} s)")
