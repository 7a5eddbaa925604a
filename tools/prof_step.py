"""cProfile of one fit+update+predict step (host side) for a bench config (diagnostic)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import deepstructuredmixtures_amd as dsm
cfg = sys.argv[1] if len(sys.argv) > 1 else "dsmgp_n100k_d8_depth4"
model, X, y, Xt, ptr, idx = bench.build_model(cfg, 0, 1, 0)
def step():
    dsm.fit(model); dsm.update(model); return dsm.predict(model, Xt)
step(); step()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
