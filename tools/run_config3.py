"""BASELINE configs 2 and 3 timed on one GPU: single exact GP N=4096 D=4 IsoSE, and buildPoE K=8 M=200 N=50k D=8 ArdSE
(128 independent experts).  fit! + predict per step after warm-up (diagnostic; parity is in tests/test_gpu_parity.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepstructuredmixtures_amd as dsm

def timeit(f, n=5):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n


def spread(f, n=30):
    """Per-call wall times of n calls after two warm-up calls: (min, median, max, index of the slowest call)."""
    f(); f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts)
    return ts.min(), float(np.median(ts)), ts.max(), int(ts.argmax()), ts

X, y, Xt = dsm.regression_data(4096, 4, seed=20202)
gp = dsm.GaussianProcess(X, y, kernel=dsm.IsoSE(np.log(0.3), 0.0), logNoise=np.log(0.1))
xa = os.environ.get("DSMGP_RUN_ARGS", "")      # A/B runs (tools/ab_libs.sh): the same switches as bench.py
def options(ctx):
    if "--no-diag-ahead" in xa:
        ctx.set_option(dsm.hipabi.OPT_DIAG_IN_UPDATE, 0)
    if "--graph" in xa:
        ctx.set_option(dsm.hipabi.OPT_FIT_GRAPH, 1)
    if "--lanes" in xa:
        ctx.set_option(dsm.hipabi.OPT_LANES, int(xa.split("--lanes")[1].split()[0]))
options(gp.model.ctx)
def step2():
    dsm.update_cholesky(gp); return dsm.prediction(gp, Xt)
lo, med, hi, _, _ = spread(step2)
print(f"config 2 (single GP n=4096, n_t={Xt.shape[0]}): {med * 1e3:.2f} ms per update_cholesky! + prediction (median of 30; min {lo * 1e3:.2f}, max {hi * 1e3:.2f})")

X, y, Xt = dsm.regression_data(50_000, 8, seed=20203)
m = dsm.buildPoE(X, y, 8, M=200, kernel=dsm.ArdSE(np.log(np.full(8, 0.3)), 0.0), logNoise=np.log(0.1),
                 meanFun=dsm.ConstMean(float(np.mean(y))), seed=20203)
n = np.array([lf.nobs for lf in m.leaves])
options(m.ctx)
def step3():
    dsm.fit(m); return dsm.predict(m, Xt)
import gc
phase = {}
def step3():
    t0 = time.perf_counter(); dsm.fit(m); t1 = time.perf_counter(); r = dsm.predict(m, Xt); t2 = time.perf_counter()
    phase["fit"], phase["predict"], phase["gc"] = t1 - t0, t2 - t1, gc.get_count()
    return r
calls = []
def step3_logged():
    r = step3()
    calls.append(dict(phase))
    return r
lo, t, hi, worst, ts = spread(step3_logged)
w = calls[2 + worst]
print(f"config 3 per-call spread over 30 calls: min {lo * 1e3:.2f} median {t * 1e3:.2f} max {hi * 1e3:.2f} ms; slowest call #{worst}: fit "
      f"{w['fit'] * 1e3:.2f} ms, predict {w['predict'] * 1e3:.2f} ms, gc counts {w['gc']}; calls above 1.5 x median: "
      f"{[(i, round(v * 1e3, 1)) for i, v in enumerate(ts) if v > 1.5 * t]}")
m.ctx.set_profile(2); step3(); tm = m.ctx.timings()
print(f"config 3 (PoE ArdSE, {m.L} experts n={n.min()}..{n.max()}, n_t={Xt.shape[0]} x {m.L}): {t * 1e3:.2f} ms per fit! + predict; "
      f"device: " + ", ".join(f"{k} {v * 1e3:.2f}" for k, v in tm.items() if v > 5e-5))
