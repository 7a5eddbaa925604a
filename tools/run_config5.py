"""BASELINE config 5 at full size on ONE GPU through the streaming (factor-and-discard) context:
buildDSMGP K=4 splits, V=3 sum children, M=500, N=500k, D=16, kernels [IsoSE, IsoLinear], depth 2 -> 288 leaf GPs,
n up to ~89k (a 64 GB factor), 2.8 TB of factors in total.  One fit! + update! + predict, optionally one train! iteration.

    python tools/run_config5.py [--train] [--N 500000]
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepstructuredmixtures_amd as dsm

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=500_000)
ap.add_argument("--D", type=int, default=16)
ap.add_argument("--M", type=int, default=500)
ap.add_argument("--train", action="store_true")
ap.add_argument("--host-only", action="store_true", help="build the tree and schedule only (no GPU)")
args = ap.parse_args()

t0 = time.perf_counter()
X, y, Xt = dsm.regression_data(args.N, args.D, seed=20205)
kern = [dsm.IsoSE(np.log(0.3), 0.0), dsm.IsoLinear(np.log(1.5))]
model = dsm.buildDSMGP(X, y, 3, 4, M=args.M, D=2, kernel=kern, logNoise=np.log(0.1), seed=20205, fit_now=False,
                       stream_budget=None if args.host_only else "auto")
n = np.array([lf.nobs for lf in model.leaves], dtype=np.float64)
print(f"# built in {time.perf_counter() - t0:.1f} s: {model.L} leaves, n = {int(n.min())}..{int(n.max())}, "
      f"factors {np.sum(n * n) * 8 / 1e12:.2f} TB, Cholesky flops {np.sum(n ** 3) / 3:.3e}", flush=True)
if args.host_only:
    sys.exit(0)
out = {"config": f"N={args.N} D={args.D} M={args.M} depth 2, [IsoSE, IsoLinear], {model.L} leaves, n max {int(n.max())}"}
if args.train:                        # one train! iteration: fit + gradients per leaf group, ADAM step, final fit
    t0 = time.perf_counter()
    dsm.train(model, dsm.ADAM(), iterations=1)
    out["train_1_iteration_plus_final_fit_s"] = time.perf_counter() - t0
    print(f"# train: {out['train_1_iteration_plus_final_fit_s']:.1f} s", flush=True)
    out["passes"] = getattr(model.ctx, "passes", None)
    g_ = getattr(model.ctx, "groups", None)     # (train() restores the plain streaming context when it is done)
    out["groups"] = [int(len(g)) for g in g_] if g_ is not None else None
    out["root_mll"] = dsm.update(model)
    print(json.dumps(out))
    sys.exit(0)
dsm.resident_test(model, Xt)          # the test rows ride through the factorisation of every leaf group
model.ctx.set_profile(1)
t0 = time.perf_counter(); dsm.fit(model); out["fit_s"] = time.perf_counter() - t0
ctx = model.ctx
tm = ctx.timings()
out["device_fit_s"] = tm.get("total_fit", 0.0)
out["groups"] = [int(len(g)) for g in ctx.groups] if getattr(ctx, "groups", None) is not None else None
out["update_tflops"] = ctx.work()[0] / max(tm.get("chol_update", 0.0), 1e-9) / 1e12 if tm.get("chol_update", 0.0) > 0 else None
out["host_wall_by_call"] = {k: round(v, 2) for k, v in getattr(ctx, "host_seconds", {}).items()}
print(f"# fit {out['fit_s']:.1f} s (device {out['device_fit_s']:.1f} s, groups {out['groups']})", flush=True)
t0 = time.perf_counter(); z = dsm.update(model); mu, var = dsm.predict(model, Xt); out["update_predict_s"] = time.perf_counter() - t0
out["root_mll"] = z
out["rmse"] = float(np.sqrt(np.mean((mu - np.mean(y)) ** 2)))
out["finite"] = bool(np.all(np.isfinite(mu)) and np.all(var > 0))
out["cholesky_tflops"] = float(np.sum(n ** 3) / 3 / out["fit_s"] / 1e12)
print(json.dumps(out))
