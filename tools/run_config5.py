"""BASELINE config 5 at full size on ONE GPU through the streaming (factor-and-discard) context:
buildDSMGP K=4 splits, V=3 sum children, M=500, N=500k, D=16, kernels [IsoSE, IsoLinear], depth 2 -> 288 leaf GPs,
n up to ~89k (a 64 GB factor), 2.8 TB of factors in total.  One fit! + update! + predict, optionally one train! iteration.

    python tools/run_config5.py [--train] [--N 500000]

A run of gpurun is limited to 20 minutes and a train! iteration of this model takes 231 s on one GPU, so the 20 iterations
SURVEY 8(d) names are run as five calls of four: --save-hyper writes the hyper-vector after every optimiser step, --load-hyper
starts the next call from it (exact: the reference's ADAM carries no state from one iteration to the next).
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
ap.add_argument("--iterations", type=int, default=1, help="train! iterations to time (with --train)")
ap.add_argument("--host-only", action="store_true", help="build the tree and schedule only (no GPU)")
ap.add_argument("--load-hyper", default=None, help=".npy hyper-vector to start train! from (instead of the random initialisation)")
ap.add_argument("--save-hyper", default=None, help=".npy file that receives the hyper-vector after every optimiser step")
ap.add_argument("--lanes", type=int, default=None, help="DSMGP_OPT_LANES inside every leaf group (default: automatic = one under the pool)")
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
if args.train:                        # train! iterations (src/optimisers.jl:40-80): fit + gradients per leaf group, ADAM step; then the final fit
    class TimedADAM(dsm.ADAM):        # the optimiser is applied once per iteration: its calls are the iteration boundaries
        stamps = []

        def apply(self, hyp, g):
            TimedADAM.stamps.append(time.perf_counter())
            print(f"# iteration {len(TimedADAM.stamps)} done at {TimedADAM.stamps[-1] - t0:.1f} s", flush=True)
            step = super().apply(hyp, g)
            if args.save_hyper:
                np.save(args.save_hyper, hyp + step)
            return step
    if args.load_hyper:
        dsm.setparams(model, np.load(args.load_hyper))
    t0 = time.perf_counter()
    _, hist = dsm.train(model, TimedADAM(), iterations=args.iterations, earlystop=10 ** 9, randinit=not args.load_hyper)
    t_end = time.perf_counter()
    st = [t0] + TimedADAM.stamps
    out["train_iteration_s"] = [round(b - a, 2) for a, b in zip(st[:-1], st[1:])]
    out["final_fit_s"] = round(t_end - st[-1], 2)
    out["train_total_s"] = round(t_end - t0, 2)
    out["mll_history"] = [float(v) for v in hist]
    out["hyper_after"] = [float(v) for v in dsm.getparams(model)]
    print(f"# train: iterations {out['train_iteration_s']} s, final fit {out['final_fit_s']} s", flush=True)
    out["passes"] = getattr(model.ctx, "passes", None)
    g_ = getattr(model.ctx, "groups", None)     # (train() restores the plain streaming context when it is done)
    out["groups"] = [int(len(g)) for g in g_] if g_ is not None else None
    out["root_mll"] = dsm.update(model)
    print(json.dumps(out))
    sys.exit(0)
if args.lanes is not None:
    model.ctx.set_option(dsm.hipabi.OPT_LANES, args.lanes)
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
# the algorithmic flops of the whole step: n^3/3 per factorisation + n^2 n_t per leaf for the prediction solves
from deepstructuredmixtures_amd import tree as ptree
nt = np.diff(ptree.route(model.root, Xt)[0]).astype(np.float64)
out["flops_cholesky"], out["flops_predict_solves"] = float(np.sum(n ** 3) / 3), float(np.sum(n * n * nt))
out["matrix_tflops_fit_predict"] = (out["flops_cholesky"] + out["flops_predict_solves"]) / (out["fit_s"] + out["update_predict_s"]) / 1e12
out["whole_run_frac_of_78.6"] = out["matrix_tflops_fit_predict"] / 78.6
out["lanes"] = args.lanes
print(json.dumps(out))
