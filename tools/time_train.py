"""Seconds per train! iteration on the headline model (fit + gradients + host update), and the device time of
the gradient pass alone (diagnostic; the bench metric is fit!+predict)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import deepstructuredmixtures_amd as dsm

cfg = sys.argv[1] if len(sys.argv) > 1 else "dsmgp_n100k_d8"
model, X, y, Xt, ptr, idx = bench.build_model(cfg, 0, 1, 0)
ctx = model.ctx
ctx.set_profile(2)
dsm.fit(model)
t0 = time.perf_counter(); dsm.fit(model); t_fit = time.perf_counter() - t0
t0 = time.perf_counter(); dsm.updategradients(model); t_g = time.perf_counter() - t0
tm = ctx.timings()
print(f"fit {t_fit:.4f} s   updategradients {t_g:.4f} s (device 'gradients' {tm.get('gradients', 0):.4f} s)")
t0 = time.perf_counter()
dsm.train(model, dsm.ADAM(), iterations=3)
print(f"train: {(time.perf_counter() - t0) / 3:.4f} s per iteration")
