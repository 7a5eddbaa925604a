"""predict(model, x) on rows the model has not seen, phase by phase (routing, registration, sweep, aggregation):
    [DSMGP_HOSTLOG=1] python tools/time_predict_new_rows.py [config]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import deepstructuredmixtures_amd as dsm
from deepstructuredmixtures_amd import tree as ptree, model as pmodel

cfg = sys.argv[1] if len(sys.argv) > 1 else "dsmgp_n100k_d8_depth4"
model, X, y, Xt, ptr, idx = bench.build_model(cfg, 0, 1, 0)
dsm.fit(model)
dsm.update(model)
dsm.predict(model, Xt)                                     # first use: arenas, Dinv completion
ctx = model.ctx
acc = {}


def timed(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, g)


timed(pmodel, "_routing", "routing (host: hash + route + local CSR)")
for nm in ("set_test", "set_test_routed", "predict_run", "aggregate", "set_tree"):
    if hasattr(ctx, nm):
        timed(ctx, nm, "ctx." + nm)
for rep in range(6):
    x = np.ascontiguousarray(Xt[::-1] if rep % 2 == 0 else Xt)
    acc.clear()
    t2 = time.perf_counter()
    mu, var = dsm.predict(model, x)
    t3 = time.perf_counter()
    print(f"{cfg}: predict on new rows {t3 - t2:.4f} s (device sweep {model.last_predict_seconds:.4f}): " +
          ", ".join(f"{k} {v:.4f}" for k, v in acc.items()), flush=True)

# Where does a sweep right after a registration spend the milliseconds its event span does not show?  The same call with a pause
# between registration and sweep:
if os.environ.get("DSMGP_PAUSE"):
    orig = ctx.set_test_routed

    def paused(xt):
        r = orig(xt)
        time.sleep(float(os.environ["DSMGP_PAUSE"]))
        return r
    ctx.set_test_routed = paused
    for rep in range(4):
        x = np.ascontiguousarray(Xt[::-1] if rep % 2 == 0 else Xt)
        acc.clear()
        mu, var = dsm.predict(model, x)
        print(f"{cfg}: with a {os.environ['DSMGP_PAUSE']} s pause after the registration: device sweep {model.last_predict_seconds:.4f}: " +
              ", ".join(f"{k} {v:.4f}" for k, v in acc.items()), flush=True)

# Is the device simply not available for a while after a registration?  A 1 ms one-wave kernel on the side stream, timed on the
# host right behind the registration, then the sweep.
if os.environ.get("DSMGP_PROBE_AFTER"):
    orig2 = ctx.set_test_routed

    def probed(xt):
        r = orig2(xt)
        t0 = time.perf_counter()
        ctx.clock_sample_start(1.0)
        ghz, ms = ctx.clock_sample_read()
        print(f"   1 ms sampler right behind the registration: {1e3 * (time.perf_counter() - t0):.2f} ms on the host, {ghz:.3f} GHz over {ms:.2f} ms", flush=True)
        return r
    ctx.set_test_routed = probed
    for rep in range(4):
        x = np.ascontiguousarray(Xt[::-1] if rep % 2 == 0 else Xt)
        acc.clear()
        mu, var = dsm.predict(model, x)
        print(f"{cfg}: device sweep {model.last_predict_seconds:.4f}: " + ", ".join(f"{k} {v:.4f}" for k, v in acc.items()), flush=True)
