"""Headline benchmark: fit! + predict wall-clock of a DSMGP (BASELINE.json metric) on N GPUs of one node.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher: starts its own N ranks under torch.distributed.run as a child process,
                                         before anything in this process touches the GPU, and exits with the child's code)

Workload (BASELINE.json configs[3], fits one GPU): buildDSMGP K=4 splits, V=3 sum children, M=200,
N=100k, D=8, IsoSE, tree depth 2 (reference default) -> 144 leaf GPs, n ~ 1.5k-14k; n_t = N/10 test rows.
One step = fit! (Gram assembly + batched Cholesky + forward substitution z = L^-1 (y - m) + per-leaf mll from z.z) +
update! + predict (K_tn assembly, triangular solves, predictive moments from V^T z, sum/product aggregation), inputs
resident in HBM; alpha = L^-T z is materialised on first use (gradients, download), not by fit!.
With N > 1 the SAME model is sharded leaf-wise over the ranks (strong scaling); the only exchange is
an all-gather of per-leaf log-marginals after fit! and of the aggregation's partial sums (3 n_t doubles per rank) after
predict (RCCL over xGMI).  Rank 0 prints one JSON line.

    python bench.py --mode train [--steps 5]     one train! iteration (src/optimisers.jl:40-80) per step: setparams!,
                                                 fit!, mll, updategradients!, grad of the tree mll, stateless-ADAM step
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC_FILE = "r06_update_kernel_traffic.json"


def source_stamp():
    """sha256 over the sources that decide what the timed kernels do (bench.py, the HIP sources and the header): the PMC
    traffic figure under profiles/ carries the stamp of the tree it was measured on, and is only quoted while it matches
    the tree that runs (there is no .git on the GPU box to ask for a commit)."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "include", "dsmgp_hip.h")]
    csrc = os.path.join(ROOT, "deepstructuredmixtures_amd", "csrc")
    files += sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hpp", ".cpp", ".sh")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
F64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X public spec, fp64 matrix = fp64 vector; the in-container guide lists no fp64 figure

CONFIGS = {
    # name: (N, D, K_sum_children, V_splits, M, depth, kernel)
    "dsmgp_n100k_d8": dict(N=100_000, D=8, K=3, V=4, M=200, depth=2),
    "dsmgp_n20k_d8": dict(N=20_000, D=8, K=3, V=4, M=200, depth=2),
    "dsmgp_n100k_d8_depth3": dict(N=100_000, D=8, K=3, V=4, M=200, depth=3),
    "dsmgp_n100k_d8_depth4": dict(N=100_000, D=8, K=3, V=4, M=200, depth=4),   # 18,461 leaves, n = 103..2668
    # BASELINE config 5: kernel vector [IsoSE, IsoLinear], 288 leaves up to n = 83k, 2.8 TB of factors: every rank
    # streams its leaf groups through HBM (hipabi.StreamingContext); ~112 s per step on one GPU
    "dsmgp_n500k_d16_kvec": dict(N=500_000, D=16, K=3, V=4, M=500, depth=2, kvec=True, stream="auto"),
}


# hyper-parameters of the timing runs: "survey" = IsoSE(log l = log 0.3, log s = 0), logNoise = log 0.1 -- the
# conditioning a fitted model has; "reference-default" = the reference's own defaults IsoSE(1.0, 1.0), logNoise = 1.0
# (src/treeStructure.jl:332-334: log scale, i.e. l = s = e, noise variance e^2 = 7.4: a trivially well-conditioned Gram matrix;
# the secondary point of SURVEY 8(d) -- same launches, same flops, other numbers in the tiles)
HYPER = {"survey": dict(logl=float(np.log(0.3)), logs=0.0, lognoise=float(np.log(0.1))),
         "reference-default": dict(logl=1.0, logs=1.0, lognoise=1.0)}


def build_model(cfg, rank, world, device, n_sub=1, hyper="survey"):
    import deepstructuredmixtures_amd as dsm
    from deepstructuredmixtures_amd import dist as pdist, tree as ptree
    c = CONFIGS[cfg]
    h = HYPER[hyper]
    X, y, Xt = dsm.regression_data(c["N"], c["D"], seed=20204)
    kern = dsm.IsoSE(h["logl"], h["logs"])
    if c.get("kvec"):
        kern = [kern, dsm.IsoLinear(np.log(1.5))]
    t0 = time.perf_counter()
    model = dsm.buildDSMGP(X, y, c["K"], c["V"], M=c["M"], D=c["depth"], kernel=kern,
                           logNoise=h["lognoise"], seed=20204, fit_now=False, device=device, n_sub=n_sub,
                           stream_budget=c.get("stream"))
    model.build_seconds = time.perf_counter() - t0      # host only: tree + overlap (no device call yet)
    ptr, idx = ptree.route(model.root, Xt)
    if world > 1:
        op, src, _ = ptree.share_schedule(model.leaves, model.D, 0.05)
        model.shard = pdist.Shard.lpt([lf.nobs for lf in model.leaves], op, src, rank, world, n_test=np.diff(ptr))
    return model, X, y, Xt, ptr, idx


def default_sub(world):
    """Concurrent contexts per GPU (hipabi.MultiContext: the leaves of this rank split over several contexts driven from
    host threads, so that one context's dependent chain of diagonal block / panel solve / reduce launches runs under
    another's update launches).  Since round 5 the library does that INSIDE one context (two leaf lanes, DSMGP_OPT_LANES,
    --lanes: no second copy of X, no host threads, the device exchange stays possible), so this stays at ONE.  History:
    ONE since round 3, at every world size: with the chain itself shortened (diagonal-block
    launch 51 -> 35 us, fused shallow steps) extra contexts no longer pay on the shards of a multi-GPU job -- one shard at
    a time on one GPU, same box, 1 / 2 / 3 contexts: 2 ranks 0.1996 / 0.1979 / 0.2054 s, 4 ranks 0.1051 / 0.1067 / 0.1142
    and 0.1062 / 0.1072 / 0.1144, 8 ranks 0.0582 / 0.0638 / 0.0646 and 0.0593 / 0.0581 / 0.0635 (round 2 ran 8 ranks with
    3, 4 and 2 ranks with 2 contexts: 0.0651 -> 0.0614 s then) -- and with one context per GPU the exchange of a `nccl` job
    stays on the device (dist.Shard.device_comm).  On a full GPU two contexts measure -1.4 % on the headline step and
    -2.7 % at depth 4 (profiles/r04_DESIGN_history.md 8d); N = 1 stays at one too: with two contexts' kernels sharing the chip the per-launch
    duration behind `roofline` is no longer a property of the kernel.  --sub N overrides."""
    return 1


def best_blas_threads():
    """The BLAS thread count at which this host's LAPACK runs a dpotrf of order 4096 fastest, among 8 .. the count the library
    picked by itself (OpenBLAS starts 64 threads on the 256-CPU hosts of the GPU boxes, whose share of one GPU job is smaller:
    oversubscribed, it factorises at a few tens of GFLOP/s) -> (threads, {threads: GFLOP/s}).  The baseline then runs at its best."""
    try:
        import threadpoolctl
    except Exception:
        return None, {}
    top = blas_threads() or 1
    try:
        top = min(top, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    cand = sorted({t for t in (8, 16, 32, 64, top) if 1 <= t <= max(top, 1)} | {top})
    sweep = {}
    for t in cand:
        with threadpoolctl.threadpool_limits(limits=t, user_api="blas"):
            sweep[t] = round(dpotrf_probe(4096)["gflops"], 1)
    return max(sweep, key=sweep.get), sweep


def cpu_baseline(model, X, y, Xt, ptr, idx, budget_s=20.0):
    best_t, sweep = best_blas_threads()
    if best_t is None:
        return _cpu_baseline(model, X, y, Xt, ptr, idx, budget_s, None, sweep)
    import threadpoolctl
    with threadpoolctl.threadpool_limits(limits=best_t, user_api="blas"):
        return _cpu_baseline(model, X, y, Xt, ptr, idx, budget_s, best_t, sweep)


def _cpu_baseline(model, X, y, Xt, ptr, idx, budget_s, best_t, sweep):
    """Oracle ("port" of the reference's per-leaf arithmetic, LAPACK via SciPy) on a bounded sample of
    the same workload: whole leaves, lean form (one potrf per leaf, diag-only variance), timed on the
    host cores; every other leaf is priced with the cost model n^3/3 + n^2 (n_t + 2) at the measured rate interpolated
    over the leaf size.  The sample always holds
    the largest and the smallest leaf; one mid-size leaf is held out of the scaling and predicted by it (the model's
    extrapolation error is part of the result); "as written" is timed on eight leaves."""
    from oracle import gp as ogp
    nobs = np.array([lf.nobs for lf in model.leaves], dtype=np.float64)
    nt = np.diff(ptr).astype(np.float64)
    cost = nobs ** 3 / 3 + nobs ** 2 * (nt + 2)
    order = np.argsort(nobs)

    def make(j):
        lf = model.leaves[j]
        return ogp.GaussianProcess(X[lf.obs], y[lf.obs], lf.mean.m, ogp.make_kernel(lf.kernel.kind, lf.kernel.loghyp()),
                                   lf.logNoise, exact_dist=False)

    def lean(j):
        t0 = time.perf_counter()
        g = make(j).update_cholesky()
        g.mll()
        rows = idx[ptr[j]:ptr[j + 1]]
        if rows.size:
            g.prediction(Xt[rows])
        return time.perf_counter() - t0

    lean(int(order[0]))                                     # untimed: BLAS thread pool start-up, page faults
    held = int(order[len(order) // 2])                      # held out: predicted by the scaling, then measured
    times = {}
    spent = 0.0
    picks = [int(order[-1]), int(order[0])]                 # the largest leaf (3 % of the flops at the headline config) first
    for q in (16, 8, 4, 2):
        picks += [int(j) for j in order[q // 2::q]]
    for j in picks:
        if j in times or j == held:
            continue
        times[j] = lean(j)
        spent += times[j]
        if spent > budget_s:
            break
    sample = list(times)
    # extrapolation: the measured rate (model flops per second) interpolated over the leaf size -- LAPACK runs the small
    # leaves at a tenth of the rate of the large ones, so one pooled rate would be wrong for both
    by_n = sorted(sample, key=lambda j: nobs[j])
    rate = lambda n: np.interp(n, [nobs[j] for j in by_n], [cost[j] / times[j] for j in by_n])  # noqa: E731
    est = float(np.sum(cost / rate(nobs)))
    t_held = lean(held)
    pred_held = float(cost[held] / rate(nobs[held]))
    threads = best_t or blas_threads() or os.cpu_count() or 1
    # "as written" (SURVEY 8(d)): the reference factorises every leaf twice per fit! (F3), forms the full K_tt and
    # V^T V in prediction and predicts in two passes (F10).  Timed on up to eight sampled leaves SPREAD OVER THE SIZE RANGE
    # (quantiles of the sample by n, the smallest and the largest included: LAPACK runs the small leaves at a tenth of its
    # large-leaf rate, so a ratio taken on the cheapest leaves alone says little about the leaves that carry the estimate)
    # against the lean form of the same leaves; the ratio of the two sums -- cost-weighted by construction -- scales the lean
    # estimate.  Informative only -- the lean figure is the baseline.
    lean8 = written8 = 0.0
    by_n = sorted(sample, key=lambda j: nobs[j])
    eight = [by_n[i] for i in sorted({int(round(q)) for q in np.linspace(0, len(by_n) - 1, min(8, len(by_n)))})]
    for j in eight:
        rows = idx[ptr[j]:ptr[j + 1]]
        lean8 += times[j]
        t0 = time.perf_counter()
        g = make(j).update_cholesky().update_cholesky()
        g.mll()
        g.prediction(Xt[rows], full_cov=True)
        g.prediction(Xt[rows], full_cov=True)
        written8 += time.perf_counter() - t0
    by_size = sorted(sample, key=lambda j: nobs[j])
    return {"value": est, "unit": "s", "cores": int(threads), "kind": "port",
            "blas_threads": int(threads), "blas_thread_sweep_dpotrf4096_gflops": sweep, "host_cpus": os.cpu_count(),
            "dpotrf_probe": dpotrf_probe(),      # how strong this baseline's own engine is on this host (a weakly threaded
                                                 # OpenBLAS makes the GPU/CPU ratio large: the ratio is no credit, the roofline is)
            "sample": f"{len(sample)} of {model.L} leaves (n={int(nobs[sample].min())}..{int(nobs[sample].max())}, the largest leaf "
                      f"included) timed {spent:.1f} s with the NumPy/LAPACK oracle, one potrf per leaf + alpha + diag-only predict; "
                      f"every other leaf priced at n^3/3 + n^2(n_t+2) flops over the measured rate interpolated at its size",
            "per_leaf_gflops": [{"n": int(nobs[j]), "seconds": round(times[j], 4), "gflops": round(cost[j] / times[j] / 1e9, 1)}
                                for j in by_size],
            "held_out_leaf": {"n": int(nobs[held]), "measured_s": t_held, "predicted_s": pred_held,
                              "relative_error": (pred_held - t_held) / t_held},
            "as_written_value": est * written8 / lean8,
            "as_written_note": f"two factorisations per leaf, full predictive covariance, two predict passes: "
                               f"{written8 / lean8:.2f}x the lean form on {len(eight)} sampled leaves spread over n = "
                               f"{int(nobs[eight[0]])}..{int(nobs[eight[-1]])} "
                               f"({lean8:.1f} s vs {written8:.1f} s)"}


def dpotrf_probe(n=8192):
    """GFLOP/s of ONE LAPACK dpotrf of order n on this host with the BLAS threads the oracle runs on: how strong the CPU baseline's
    own engine is (SciPy's OpenBLAS reaches a few hundred GFLOP/s on a well-threaded 64-core box, tens when it is not)."""
    import scipy.linalg as sl
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, 256))
    S = A @ A.T
    S[np.diag_indices(n)] += float(n)
    sl.lapack.dpotrf(S[:512, :512].copy(order="F"), lower=1)       # thread pool start-up
    F = np.asfortranarray(S)
    t0 = time.perf_counter()
    _, info = sl.lapack.dpotrf(F, lower=1, overwrite_a=1)
    dt = time.perf_counter() - t0
    return {"n": n, "seconds": dt, "gflops": n ** 3 / 3 / dt / 1e9, "info": int(info)}


def blas_threads():
    try:
        import threadpoolctl
        return max([p["num_threads"] for p in threadpoolctl.threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        return None


def short_series(ctx, step, flops_step, steps, warmup, torch):
    """A short timed series of one of the other single-GPU BASELINE configs (run by the default bench after the headline series,
    outside its timed region): `steps` calls of step() on wall clock with totals-only timing, then the same number with HIP events
    around the launches of the dominant kernel -> {step_s, roofline}.  flops_step = sum n^3/3 + n^2 n_t over the leaves."""
    ctx.set_profile(0)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    walls = []
    for _ in range(steps):
        t0 = time.perf_counter()
        step()
        walls.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    ctx.set_profile(3)      # events around the dominant launches, under the kernel names of untimed launches: a profiler's
    step()                  # average of the headline's timed instantiation (<false, 0, *>) stays the headline's
    cats, walls_p, n_upd, n_fus = {}, [], 0, 0
    for _ in range(steps):
        t0 = time.perf_counter()
        step()
        walls_p.append(time.perf_counter() - t0)
        for k, v in ctx.timings().items():
            cats[k] = cats.get(k, 0.0) + v
        n_upd += ctx.work()[1]
        n_fus += ctx.work_fused()[1]
    ctx.set_profile(0)
    fused = cats.get("chol_fused", 0.0) > cats.get("chol_update", 0.0)
    fl, n_l = (ctx.work_fused()[0], n_fus) if fused else (ctx.work()[0], n_upd)
    t_sum = cats.get("chol_fused" if fused else "chol_update", 0.0)
    t_union = cats.get("chol_fused_union" if fused else "chol_update_union", 0.0) or t_sum
    med = float(np.median(walls))
    out = {"step_s": spread(walls), "step_s_with_launch_events": spread(walls_p), "steps": steps,
           "matrix_flops_per_step": flops_step, "whole_step_tflops": flops_step / med / 1e12,
           "whole_step_frac": flops_step / med / 1e12 / F64_MATRIX_PEAK_TFLOPS,
           "device_seconds_per_step": {k: v / steps for k, v in cats.items() if v > 0}}
    if t_union > 0 and n_l > 0:
        ach = fl * steps / t_union / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "tile_fused8_kernel<0> (fused block steps: update + solve)" if fused
                           else "tile_gemm_kernel_v2<false, 0, *> (update launches)",
                           "alg_flops": fl, "achieved": ach, "peak": F64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / F64_MATRIX_PEAK_TFLOPS, "launches_per_step": n_l // steps,
                           "avg_launch_ms": t_sum / n_l * 1e3, "launch_seconds_per_step": {"sum": t_sum / steps, "union": t_union / steps},
                           "traffic": None}
    return out


def other_configs(device, torch):
    """BASELINE configs 2, 3 and 4'' (depth 4) on this GPU, each with its own model and context, after the headline series:
    {name: {step_s, roofline{kernel, alg_flops, achieved, frac}, ...}} (VERDICT r5 #1b: every single-GPU config driver-observed).
    One config failing (memory, a HIP error) leaves the others in the line."""
    import deepstructuredmixtures_amd as dsm
    from deepstructuredmixtures_amd import tree as ptree
    h = HYPER["survey"]

    def single_gp():
        # config 2: one exact GP, N = 4096, D = 4, IsoSE: update_cholesky! + prediction on N/10 rows (src/gaussianprocess.jl:82-137)
        X, y, Xt = dsm.regression_data(4096, 4, seed=20202)
        gp = dsm.GaussianProcess(X, y, kernel=dsm.IsoSE(h["logl"], h["logs"]), logNoise=h["lognoise"], device=device)
        try:
            def step():
                dsm.update_cholesky(gp)
                return dsm.prediction(gp, Xt)
            r = short_series(gp.model.ctx, step, 4096.0 ** 3 / 3 + 4096.0 ** 2 * Xt.shape[0], 30, 3, torch)
        finally:
            gp.model.ctx.close()
        r["workload"] = (f"single GaussianProcess N=4096 D=4 IsoSE: update_cholesky! + prediction(gp, x) on {Xt.shape[0]} rows; "
                         "32 dependent block steps")
        return r

    def poe_ardse():
        # config 3: buildPoE K = 8, M = 200, N = 50k, D = 8, ArdSE (additive): 128 independent experts n ~ 391
        X, y, Xt = dsm.regression_data(50_000, 8, seed=20203)
        m = dsm.buildPoE(X, y, 8, M=200, kernel=dsm.ArdSE(np.full(8, h["logl"]), h["logs"]), logNoise=h["lognoise"],
                         meanFun=dsm.ConstMean(float(np.mean(y))), seed=20203, fit_now=False, device=device)
        nobs = np.array([lf.nobs for lf in m.leaves], dtype=np.float64)
        try:
            def step():
                dsm.fit(m)
                return dsm.predict(m, Xt)
            r = short_series(m.ctx, step, float(np.sum(nobs ** 3) / 3 + np.sum(nobs ** 2) * Xt.shape[0]), 30, 3, torch)
        finally:
            m.ctx.close()
        r["workload"] = (f"buildPoE K=8 M=200 N=50000 D=8 ArdSE: {m.L} experts n={int(nobs.min())}..{int(nobs.max())}, every expert "
                         f"predicts all {Xt.shape[0]} rows; fit! + predict; the kernel function (8 exp per entry) shares the f64 pipe "
                         "with the MFMAs")
        return r

    def depth4():
        # config 4'': the headline data at tree depth 4 (18,461 leaves, n = 103..2,668: the small-leaf regime)
        c = CONFIGS["dsmgp_n100k_d8_depth4"]
        X, y, Xt = dsm.regression_data(c["N"], c["D"], seed=20204)
        t0 = time.perf_counter()
        m = dsm.buildDSMGP(X, y, c["K"], c["V"], M=c["M"], D=c["depth"], kernel=dsm.IsoSE(h["logl"], h["logs"]),
                           logNoise=h["lognoise"], seed=20204, fit_now=False, device=device)
        t_build = time.perf_counter() - t0
        nobs = np.array([lf.nobs for lf in m.leaves], dtype=np.float64)
        ptr, _ = ptree.route(m.root, Xt)
        try:
            dsm.resident_test(m, Xt)

            def step():
                dsm.fit(m)
                dsm.update(m)
                return dsm.predict(m, Xt)
            r = short_series(m.ctx, step, float(np.sum(nobs ** 3) / 3 + np.sum(nobs ** 2 * np.diff(ptr))), 10, 2, torch)
            r["lanes"] = m.ctx.lanes()
        finally:
            m.ctx.close()
        r["workload"] = (f"buildDSMGP K=4 splits V=3 sum children M=200 N=100000 D=8 IsoSE depth 4: {m.L} leaf GPs n={int(nobs.min())}.."
                         f"{int(nobs.max())}, {Xt.shape[0]} test rows x {int(ptr[-1] // Xt.shape[0])} leaves each; fit! + update! + predict")
        r["model_build_s"] = t_build
        return r

    out = {}
    for name, fn in (("single_gp_n4096", single_gp), ("poe_ardse_n50k", poe_ardse), ("dsmgp_depth4", depth4)):
        try:
            out[name] = fn()
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out

def bench_train(args, model, X, y, rank, world, td, torch):
    """One train! iteration per step (src/optimisers.jl:40-80) on the bench config: setparams!, fit!, tree mll,
    updategradients! (L^-T by blocked triangular inversion + contraction tiles), gradient of the tree mll
    (src/optimize.jl:42-89), stateless-ADAM step.  Not the BASELINE metric: printed with its own name."""
    import deepstructuredmixtures_amd as dsm
    ctx = model.ctx
    ctx.set_joint(False)
    ctx.set_profile(1)
    opt = dsm.ADAM()
    hyp = dsm.getparams(model).copy()

    def iteration(h):
        t = {}
        a = time.perf_counter()
        dsm.setparams(model, h)
        dsm.fit(model)
        ell = dsm.mll(model)
        b = time.perf_counter()
        dsm.updategradients(model)
        c_ = time.perf_counter()
        g = dsm.grad_mll(model)
        h = h + opt.apply(h, g)
        d = time.perf_counter()
        t.update(fit_wall=b - a, grad_wall=c_ - b, tree_wall=d - c_, mll=ell)
        return h, t

    t_first = time.perf_counter()
    hyp, first = iteration(hyp)                      # builds the gradient plan and the L^-T arena
    t_first = time.perf_counter() - t_first
    for _ in range(max(0, args.warmup - 1)):
        hyp, _t = iteration(hyp)
    torch.cuda.synchronize()
    if td is not None:
        td.barrier()
    t0 = time.perf_counter()
    acc = {}
    hist = []
    for _ in range(args.steps):
        hyp, t = iteration(hyp)
        hist.append(t.pop("mll"))
        for k, v in t.items():
            acc[k] = acc.get(k, 0.0) + v
        for k, v in ctx.timings().items():
            acc["dev_" + k] = acc.get("dev_" + k, 0.0) + v
    torch.cuda.synchronize()
    if td is not None:
        td.barrier()
    elapsed = time.perf_counter() - t0
    if td is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu")
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        elapsed = float(tmax.item())
    per = elapsed / args.steps
    n = args.steps
    fi, fc, ntiles = ctx.work_gradients() if hasattr(ctx, "work_gradients") else (0.0, 0.0, 0)
    fl_upd, _ = ctx.work()
    if rank == 0:
        c = CONFIGS[args.config]
        dev = {k[4:]: v / n for k, v in acc.items() if k.startswith("dev_") and v > 0}
        out = {"metric": "train! iteration wall-clock, DSMGP N=%dk D=%d" % (c["N"] // 1000, c["D"]), "value": per, "unit": "s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": per * 1e3,
               "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"one train! iteration (setparams!, fit!, mll, updategradients!, grad of the tree mll, "
                                      f"stateless ADAM step) of buildDSMGP K=4 splits V=3 sum children M={c['M']} N={c['N']} "
                                      f"D={c['D']} depth {c['depth']}: {model.L} leaf GPs",
                          "parallelism": f"leaves sharded over {world} GPU(s)"},
               "first_iteration_s": t_first,
               "host_wall_per_iteration": {k: v / n for k, v in acc.items() if not k.startswith("dev_")},
               "device_seconds_per_iteration": dev,
               "host_overhead_per_iteration": per - dev.get("total_fit", 0.0) - dev.get("gradients", 0.0) - dev.get("alpha", 0.0),
               "mll_history": hist}
        if world == 1 and dev.get("grad_contraction", 0.0) > 0:
            ach = fc / dev["grad_contraction"] / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": F64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / F64_MATRIX_PEAK_TFLOPS, "traffic": None,
                               "kernel": f"tile_graddot_kernel: {ntiles} contraction tiles in one launch, n^3/3 flops per IsoSE leaf "
                                         "(epilogue: squared distance + exp per element)"}
            out["roofline_inverse"] = {"bound": "mfma", "achieved": fi / dev["grad_inverse"] / 1e12, "peak": F64_MATRIX_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": fi / dev["grad_inverse"] / 1e12 / F64_MATRIX_PEAK_TFLOPS,
                                       "kernel": "L^-T by blocked triangular inversion: tile_gemm_kernel_v2 + tile_trsm_kernel + "
                                                 "tile_reduce_kernel launches of dsmgp_gradients, n^3/3 flops per factor"}
            out["roofline_fit_update"] = {"achieved": fl_upd / dev["chol_update"] / 1e12 if dev.get("chol_update") else None,
                                          "unit": "TFLOP/s", "kernel": "update launches of fit! inside the iteration"}
        print(json.dumps(out))
    if td is not None:
        td.barrier()
        td.destroy_process_group()


def spread(v):
    """{min, median, max} of a series of per-step values."""
    v = np.asarray(list(v), dtype=np.float64)
    return {"min": float(v.min()), "median": float(np.median(v)), "max": float(v.max())} if v.size else None


def self_launch(args):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves, as
    `python -m torch.distributed.run ... bench.py <same arguments>` in a child process, and exit with its code.  Decided before
    this process imports torch or makes any GPU call (it never does: a process that has initialised the GPU must not exec or be
    the parent of the job's device work); rank 0 of the child prints the JSON line on the stdout we hand down."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver supports dmabuf IPC only (RCCL between processes)
    print(f"# bench.py --gpus {args.gpus} without a launcher: starting {args.gpus} ranks under torch.distributed.run "
          f"(127.0.0.1:{port})", file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def init_distributed(rank, world, local_rank, torch):
    """Process groups of an N > 1 run -> (torch.distributed, backend of the data path, local device index).

    RCCL ("nccl") is what the run is meant to use; whether it comes up is a verdict the ranks must AGREE on before any of them
    depends on it (round-4 advisor: decided per rank inside try/except, a partial failure left some ranks in nccl and the
    others in gloo until the 300 s timeout).  So: (1) the DEFAULT group is gloo -- it exists on every rank whatever the GPUs
    do, and carries the barriers, the timing reduction and the per-rank report; (2) every rank opens a nccl group BESIDE it and
    proves it with one all-reduce; (3) the verdicts are MIN- and MAX-reduced over gloo.  All yes: the exchanges of the path
    (dist.GROUP) travel over the proven nccl group -- nothing is destroyed or re-initialised.  All no (fewer GPUs than ranks, an IPC
    or driver problem: the same everywhere): they travel over gloo and the line says so (exchange_backend).  Mixed: exit 3."""
    import datetime
    import torch.distributed as td
    from deepstructuredmixtures_amd import dist as pdist
    ndev = torch.cuda.device_count()
    want = os.environ.get("DSMGP_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the N > 1 path on one GPU
    local_rank = local_rank % max(1, ndev)
    torch.cuda.set_device(local_rank)
    td.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    if want != "nccl":
        return td, "gloo", local_rank
    ok, why, g = 1, "", None
    if ndev < world:                                          # the same on every rank of the node: nobody opens a nccl group
        ok, why = 0, f"{world} ranks need {world} GPUs, {ndev} visible"
    else:
        try:
            g = td.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
            probe = torch.ones(1, device="cuda")
            td.all_reduce(probe, group=g)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"RCCL all-reduce of ones over {world} ranks gave {probe.item()}")
        except Exception as e:      # noqa: BLE001
            ok, why = 0, str(e)
    lo = torch.tensor([ok], dtype=torch.int32)
    hi = torch.tensor([ok], dtype=torch.int32)
    td.all_reduce(lo, op=td.ReduceOp.MIN)
    td.all_reduce(hi, op=td.ReduceOp.MAX)
    if int(lo.item()) != int(hi.item()):
        print(f"# rank {rank}: the ranks disagree on RCCL (here: {'ok' if ok else why}); giving up on every rank", file=sys.stderr, flush=True)
        os._exit(3)                                           # no destructor may wait for a half-built communicator
    if int(lo.item()) == 0:
        print(f"# rank {rank}: RCCL process group not usable ({why}); falling back to gloo for the exchange", file=sys.stderr)
        return td, "gloo", local_rank
    pdist.GROUP = g
    return td, "nccl", local_rank


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="dsmgp_n100k_d8", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="fit_predict", choices=["fit_predict", "train"],
                    help="fit_predict: the BASELINE metric (default); train: one train! iteration per step")
    ap.add_argument("--hyper", default="survey", choices=sorted(HYPER),
                    help="hyper-parameters of the run: survey (default: IsoSE(log 0.3, 0), logNoise log 0.1) or the "
                         "reference's own defaults IsoSE(1, 1), logNoise 1 (src/treeStructure.jl:332-334), SURVEY 8(d)'s secondary point")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short series of BASELINE configs 2, 3 and depth 4 that follow the headline series (N = 1, default config)")
    ap.add_argument("--no-profile", action="store_true", help="do not time kernel categories with hipEvents")
    ap.add_argument("--unfused-gram", action="store_true",
                    help="diagnostic: every Gram tile through memory first (DSMGP_OPT_FUSED_GRAM = 0), for A/B runs")
    ap.add_argument("--no-fused-steps", action="store_true", help="diagnostic: DSMGP_OPT_FUSED_STEPS = 0, for A/B runs")
    ap.add_argument("--graph", action="store_true", help="diagnostic: DSMGP_OPT_FIT_GRAPH = 1 (the untimed fits replay a captured hipGraph)")
    ap.add_argument("--no-diag-ahead", action="store_true",
                    help="diagnostic: DSMGP_OPT_DIAG_IN_UPDATE = 0 (a diagonal-block launch per classic step), for A/B runs")
    ap.add_argument("--lanes", type=int, default=None, choices=[0, 1, 2, 3, 4],
                    help="leaf lanes inside the context (DSMGP_OPT_LANES): default 0 = automatic (two from 8 independent sharing groups on)")
    ap.add_argument("--sub", type=int, default=None,
                    help="concurrent contexts per GPU (hipabi.MultiContext); default 1")
    ap.add_argument("--simulate-shard", default=None, metavar="R/W",
                    help="diagnostic: run only rank R's leaf shard of a W-rank job on this one GPU (no exchange; "
                         "not a valid bench line)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and not args.simulate_shard:
        self_launch(args)                       # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    import torch
    td = None
    exchange_backend = None
    if world > 1:
        # gloo announces its connections on the process's stdout ("[Gloo] Rank 0 is connected to 1 peer ranks ..."): while the
        # groups come up, file descriptor 1 points at stderr, so that rank 0's stdout carries the ONE JSON line and nothing else
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            td, exchange_backend, local_rank = init_distributed(rank, world, local_rank, torch)
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import deepstructuredmixtures_amd as dsm
    if args.simulate_shard:
        from deepstructuredmixtures_amd import dist as pdist
        r_, w_ = map(int, args.simulate_shard.split("/"))
        n_sub = args.sub if args.sub is not None else default_sub(w_)
        model, X, y, Xt, ptr, idx = build_model(args.config, r_, w_, local_rank, n_sub, args.hyper)
        own = model.shard.owner
        loc = np.flatnonzero(own == r_)

        class _Solo(pdist.Shard):   # same leaf subset, no communication: other ranks' values are placeholders
            def gather_leaf_values(self, v):
                out = np.zeros(self.owner.size)
                out[self.local] = v
                return out

            def gather_leaf_columns(self, v):
                out = np.zeros((self.owner.size, v.shape[1]))
                out[self.local] = v
                return out

            def allgather_sum(self, local):
                return np.asarray(local, dtype=np.float64).copy()

            def gather_ragged(self, flat, counts):
                out = np.ones(int(np.sum(counts)))
                p = np.concatenate([[0], np.cumsum(counts)])
                pos = 0
                for g in self.local:
                    out[p[g]:p[g + 1]] = flat[pos:pos + counts[g]]
                    pos += counts[g]
                return out
        model.shard = _Solo(own, r_, w_)
        n3 = np.array([lf.nobs for lf in model.leaves], dtype=float) ** 3
        print(f"# shard {r_}/{w_}: {loc.size} leaves, {n3[loc].sum() / n3.sum():.3f} of the Cholesky flops", file=sys.stderr)
    else:
        n_sub = args.sub if args.sub is not None else default_sub(world)
        model, X, y, Xt, ptr, idx = build_model(args.config, rank, world, local_rank, n_sub, args.hyper)
    ctx_ok, ctx_err = 1, None
    try:
        ctx = model.ctx
    except Exception as e:      # a sub-context failed to come up (several HIP contexts per GPU next to the process group)
        if n_sub <= 1:
            raise
        ctx_ok, ctx_err = 0, e
    if n_sub > 1:               # fall back to one context per GPU on EVERY rank or on none: creating the single context may
        if td is not None:      # enter a collective (the opt-in device exchange), which one rank alone must not do
            flag = torch.tensor([ctx_ok], dtype=torch.int32,
                                device="cpu")
            td.all_reduce(flag, op=td.ReduceOp.MIN)
            ctx_ok = int(flag.item())
        if not ctx_ok:
            print(f"# rank {rank}: {n_sub} contexts per GPU failed ({ctx_err or 'on another rank'}); falling back to one",
                  file=sys.stderr)
            n_sub = 1
            model._n_sub, model._ctx = 1, None
            ctx = model.ctx
    # N > 1 over RCCL: the library's own communicator (dsmgp_fit_exchange / dsmgp_aggregate_exchange: payload in HBM, collective
    # on the context's stream) is built and PROVEN now, before the first series, and left on standby: the timed series below
    # travels over torch.distributed (the path every world-2 test covers), a second series of the same steps then over the
    # communicator -- so the first hardware run with more than one rank shows both paths in one line
    if td is not None and exchange_backend == "nccl" and n_sub == 1 and isinstance(ctx, dsm.hipabi.Context) and not args.simulate_shard:
        model.shard.device_comm(ctx, force=True, standby=True)
    if args.unfused_gram:
        ctx.set_option(dsm.hipabi.OPT_FUSED_GRAM, 0)
    if args.no_fused_steps:
        ctx.set_option(dsm.hipabi.OPT_FUSED_STEPS, 0)
    if args.no_diag_ahead:
        ctx.set_option(dsm.hipabi.OPT_DIAG_IN_UPDATE, 0)
    if args.graph:
        ctx.set_option(dsm.hipabi.OPT_FIT_GRAPH, 1)
    if args.lanes is not None:
        ctx.set_option(dsm.hipabi.OPT_LANES, args.lanes)
    if args.mode == "train":
        return bench_train(args, model, X, y, rank, world, td, torch)
    ctx.set_profile(0 if args.no_profile else 1)   # timed region: events around the update launches only
    # the evaluation loop knows its test set: register it before the first fit, so that warm-up and timed steps run
    # the same launches (and a process-wide rocprofv3 --stats average of the update kernel is the timed-region average)
    dsm.resident_test(model, Xt)

    def sync_all():
        torch.cuda.synchronize()
        if td is not None:
            td.barrier()
            torch.cuda.synchronize()

    root = [0.0]

    def step():
        dsm.fit(model)
        root[0] = dsm.update(model)
        return dsm.predict(model, Xt)

    # f64-MFMA probe (register-only loop over the whole chip; also reports the shader clock it held) BEFORE warm-up and
    # IMMEDIATELY AFTER the timed loop: with the clock sampled inside the steps it tells box from code when two runs differ
    can_probe = hasattr(ctx, "probe_f64_mfma_detail") and not args.simulate_shard
    probe_before = ctx.probe_f64_mfma_detail(8) if can_probe else None
    t_warm = 0.1
    for _ in range(args.warmup):
        tw = time.perf_counter()
        step()
        t_warm = time.perf_counter() - tw
    sync_all()
    # shader clock held INSIDE the steps: a one-wave kernel on a stream of its own sleeps through the first 80 % of a step (its
    # update launches) and counts shader cycles per wall tick -- first and last five steps (dsmgp_clock_sample_*)
    can_clock = hasattr(ctx, "clock_sample_start") and not args.no_profile
    clock_ms = min(4000.0, max(1.0, 800.0 * t_warm))
    t0 = time.perf_counter()
    upd_launches, fused_launches = 0, 0
    cats = {}
    step_wall, step_upd, step_fused, step_fit_dev, step_pred_dev, step_clock = [], [], [], [], [], []
    xs0, xn0 = model.shard.exchange_seconds, model.shard.exchanges
    for i in range(args.steps):
        sample = can_clock and (i < 5 or i >= args.steps - 5)
        if sample:
            ctx.clock_sample_start(clock_ms)
        ts = time.perf_counter()
        mu, var = step()
        step_wall.append(time.perf_counter() - ts)
        tm = ctx.timings()
        for k, v in tm.items():
            cats[k] = cats.get(k, 0.0) + v
        step_upd.append(tm.get("chol_update_union", 0.0) or tm.get("chol_update", 0.0))      # time during which any such launch ran
        step_fused.append(tm.get("chol_fused_union", 0.0) or tm.get("chol_fused", 0.0))
        step_fit_dev.append(tm.get("total_fit", 0.0))
        step_pred_dev.append(tm.get("total_predict", 0.0))
        fl, nl = ctx.work()
        upd_launches += nl
        fused_launches += ctx.work_fused()[1]
        if sample:
            step_clock.append(ctx.clock_sample_read()[0])
    sync_all()
    elapsed = time.perf_counter() - t0
    probe_after = ctx.probe_f64_mfma_detail(8) if can_probe else None
    xs1, xn1 = model.shard.exchange_seconds, model.shard.exchanges
    if not args.no_profile:   # one more, untimed, step with every category timed: the breakdown printed below
        timed = dict(cats)
        ctx.set_profile(2)
        step()
        cats = {k: v * args.steps for k, v in ctx.timings().items()}
        for k in ("chol_update", "chol_fused", "chol_update_union", "chol_fused_union", "total_fit", "total_predict"):   # from the timed region itself
            cats[k] = timed.get(k, 0.0)
        ctx.set_profile(1)
    if td is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu")
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        elapsed = float(tmax.item())
    per_step = elapsed / args.steps
    assert args.simulate_shard or (np.all(np.isfinite(mu)) and np.all(var > 0))
    ranks_info = None
    if td is not None:          # what every rank did: an N-GPU line explains its own efficiency
        n3 = np.array([lf.nobs for lf in model.leaves], dtype=np.float64) ** 3
        loc = model.shard.local
        mine = {"rank": rank, "n_leaves": int(len(loc)), "cholesky_flop_share": float(n3[loc].sum() / n3.sum()),
                "largest_leaf": int(max([model.leaves[j].nobs for j in loc], default=0)),
                "step_s": spread(step_wall), "fit_device_s": spread(step_fit_dev), "predict_device_s": spread(step_pred_dev),
                "exchange_s_per_step": (xs1 - xs0) / args.steps, "exchanges_per_step": (xn1 - xn0) / args.steps}
        ranks_info = [None] * world
        td.all_gather_object(ranks_info, mine)

    # The same steps once more with the exchanges inside the library (RCCL on the context's stream, payload in HBM): every rank
    # holds the standby communicator or none does (its set-up MIN-reduces every verdict), so every rank enters this block or none
    device_series = None
    if td is not None and model.shard.standby_ctx is not None:
        mu_t, var_t = mu.copy(), var.copy()
        model.shard.activate_device_exchange()
        step()                                   # its fit_exchange is cross-checked against torch.distributed (dist.Shard.fit_exchange)
        sync_all()
        xd0, xdn0 = model.shard.exchange_seconds, model.shard.exchanges
        td0 = time.perf_counter()
        walls_d = []
        for _ in range(args.steps):
            ts = time.perf_counter()
            mu, var = step()
            walls_d.append(time.perf_counter() - ts)
        sync_all()
        el_d = time.perf_counter() - td0
        tmax = torch.tensor([el_d], dtype=torch.float64, device="cpu")
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        mine_d = {"rank": rank, "step_s": spread(walls_d), "exchange_s_per_step": (model.shard.exchange_seconds - xd0) / args.steps,
                  "exchanges_per_step": (model.shard.exchanges - xdn0) / args.steps, "exchange_in_use": model.shard.exchange,
                  "rccl_ranks_seen": model.shard.rccl_ranks_seen,
                  "max_abs_diff_vs_torch_series": [float(np.max(np.abs(mu - mu_t))), float(np.max(np.abs(var - var_t)))]}
        ranks_d = [None] * world
        td.all_gather_object(ranks_d, mine_d)
        device_series = {"value": float(tmax.item()) / args.steps, "unit": "s", "steps": args.steps, "ranks": ranks_d,
                         "note": "the same steps with the (mll, info) all-gather and the aggregation's partial sums exchanged inside the "
                                 "library (dsmgp_fit_exchange / dsmgp_aggregate_exchange over RCCL, device to device); the first exchange "
                                 "cross-checked against torch.distributed; `value` of the line is the torch.distributed series"}

    # roofline of the dominant kernel on this rank: the f64-MFMA update launches (tile_gemm_kernel_v2) -- or, where the fused
    # block steps dominate (many small leaves), the fused tile launches (tile_fused8_kernel: update + solve)
    alg_flops, _ = ctx.work()
    alg_fused, _ = ctx.work_fused()
    roof = None
    lanes = ctx.lanes() if hasattr(ctx, "lanes") else 1
    fused_dominant = cats.get("chol_fused", 0.0) > cats.get("chol_update", 0.0)
    if not args.no_profile and max(cats.get("chol_update", 0.0), cats.get("chol_fused", 0.0)) > 0:
        if fused_dominant:
            t_cat, n_l, fl_step = cats["chol_fused"], fused_launches, alg_fused
            kname = ("tile_fused8_kernel<0> (fused block steps: update of a tile from the kernel function, solve against the step's "
                     "diagonal block from the accumulators, one write; algorithmic flops = update + c_k^2 per solved row)")
        else:
            t_cat, n_l, fl_step = cats["chol_update"], upd_launches, alg_flops
            kname = ("tile_gemm_kernel_v2<false, 0, *> (update launches of the factorisation, test rows riding along, each task "
                     "evaluating the kernel function of its own tile; the same kernel under two symbol names: <false, 0, true> "
                     "in launches of at least two rounds that carry short tiles -- a leaf's last row tile -- and <false, 0, false> "
                     "in the others; a profiler's average of `that kernel` is the call-weighted mean of the two rows; "
                     "panel solves run as tile_trsm_kernel, split-K reduces as tile_reduce_kernel, the shallow block steps "
                     "(K <= 512) as diag_fused_reg_kernel + tile_fused8_kernel: all timed apart, device_seconds_per_step; the diagonal blocks of the other steps ride in these launches as DiagFinishTasks)")
        # With two leaf lanes the launches of the kernel overlap in time: `achieved` = their flops over the time during which ANY of
        # them ran (the union of the launch intervals, two events per launch: dsmgp_timings chol_*_union); avg_launch_ms stays
        # the mean duration of a launch, the figure a profiler's per-kernel average shows.  One lane: union = sum.
        t_union = cats.get("chol_fused_union" if fused_dominant else "chol_update_union", 0.0) or t_cat
        avg_launch = t_cat / max(1, n_l)
        flops_per_launch = fl_step * args.steps / max(1, n_l)
        achieved = fl_step * args.steps / t_union / 1e12
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if os.path.exists(tpath) and world == 1 and args.config == "dsmgp_n100k_d8" and not args.simulate_shard and not fused_dominant:
            # PMC counters cannot be collected inside this run: the figure is the one of the committed rocprofv3 --pmc
            # passes of this same command (tools/profile_round.sh), NOT a measurement of this run -- and it is quoted only
            # while the sources it was taken on are the sources that run now
            tj = json.load(open(tpath))
            if tj.get("source_stamp") == source_stamp():
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = (f"profiles/{TRAFFIC_FILE} (rocprofv3 --pmc passes of this command on this source tree, stamp "
                                  f"{tj['source_stamp']}, commit {tj.get('commit', '?')}; not measured in this run)")
            else:
                traffic_source = (f"profiles/{TRAFFIC_FILE} is stale: taken on source stamp {tj.get('source_stamp')}, "
                                  f"this tree is {source_stamp()}")
        roof = {"bound": "mfma", "achieved": achieved, "peak": F64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / F64_MATRIX_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": kname, "avg_launch_ms": avg_launch * 1e3, "launches_per_step": n_l // args.steps,
                "alg_flops_per_step": fl_step, "lanes": lanes, "launch_seconds_per_step": {"sum": t_cat / args.steps, "union": t_union / args.steps}}
        if probe_after is not None:
            # the same kernel against what THIS chip's matrix pipe delivered right after the timed loop (register-only f64 MFMA
            # loop, 8 waves per SIMD), and the peak rescaled to the shader clock held inside the steps: 78.6 TFLOP/s is
            # 256 CUs x 4 SIMDs x 32 flop per cycle at 2.4 GHz
            roof["frac_of_probe"] = achieved / probe_after["tflops"]
            if step_clock:
                ghz = float(np.median(step_clock))
                roof["clock_ghz_in_steps"] = ghz
                roof["frac_at_held_clock"] = achieved / (F64_MATRIX_PEAK_TFLOPS * ghz / 2.4)
    nobs = np.array([lf.nobs for lf in model.leaves], dtype=np.float64)
    matrix_flops_total = float(np.sum(nobs ** 3) / 3 + np.sum(nobs ** 2 * np.diff(ptr)))   # Cholesky + predict solves

    # The drop-in call pattern of src/common.jl:304 -- fit!(model) + update!(model) on a model whose fit does not know the test
    # set, then predict(model, x) running its own sweep -- as a second TIMED SERIES of the same --steps (one GPU only; the
    # step above registers its test set before the first fit, dsm.resident_test, which has no reference counterpart).
    standalone = None
    if world == 1 and not args.simulate_shard and not CONFIGS[args.config].get("stream"):
        ctx.set_profile(0)      # also renames the update kernel's instantiation: these launches stay out of the profiler's
                                # average of tile_gemm_kernel_v2<false, 0, *> (the kernel of the roofline block)
        ctx.set_joint(False)
        dsm.fit(model)          # untimed: the step lists of a fit without test rows are built on first use
        torch.cuda.synchronize()
        d_fit, d_pred, d_step = [], [], []
        for _ in range(args.steps):
            t0 = time.perf_counter()
            dsm.fit(model)
            dsm.update(model)
            t1 = time.perf_counter()
            mu_s, var_s = dsm.predict(model, Xt)
            t2 = time.perf_counter()
            d_fit.append(t1 - t0)
            d_pred.append(t2 - t1)
            d_step.append(t2 - t0)
        assert np.allclose(mu_s, mu, rtol=1e-9, atol=1e-11) and np.allclose(var_s, var, rtol=1e-8, atol=1e-12)
        # ... and predict on rows the model has not seen (the same rows in reverse order, then forward again, ...: every call a
        # new matrix to route, register and sweep; the answers must be the old ones, reversed where the rows were)
        Xt2 = np.ascontiguousarray(Xt[::-1])
        d_new = []
        for q in range(max(2, min(args.steps, 6))):
            xq = Xt2 if q % 2 == 0 else Xt
            t3 = time.perf_counter()
            mu_n, var_n = dsm.predict(model, xq)
            d_new.append(time.perf_counter() - t3)
            if q % 2 == 0:
                mu_n, var_n = mu_n[::-1], var_n[::-1]
            assert np.allclose(mu_n, mu, rtol=1e-9, atol=1e-11) and np.allclose(var_n, var, rtol=1e-8, atol=1e-12)
        ctx.set_joint(True)
        standalone = {"fit_s": spread(d_fit), "predict_s": spread(d_pred), "step_s": spread(d_step), "predict_new_rows_s": spread(d_new),
                      "note": f"{args.steps} timed repetitions of fit! + update! WITHOUT a resident test set, then predict(model, Xt) "
                              "running its own sweep (the reference's call pattern, src/common.jl:304); predict_new_rows: "
                              "predict(model, x) on a test matrix the model has not seen (routing, registration, sweep)"}

    # With two leaf lanes the launches of the roofline kernel overlap: the SAME step once more on one lane (untimed extra: plan and
    # test set are rebuilt for it), so that the per-launch figure of earlier rounds stays readable beside the union-based one.
    single_lane = None
    if roof is not None and lanes > 1 and standalone is not None:
        model.set_option(dsm.hipabi.OPT_LANES, 1)
        ctx.set_profile(3)      # per-launch events, under the kernel names of the untimed launches: a profiler's average of the
                                # timed instantiation stays the timed quantity
        dsm.resident_test(model, Xt)
        step()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        tm1 = ctx.timings()
        slot = "chol_fused" if fused_dominant else "chol_update"
        fl1, nl1 = ctx.work_fused() if fused_dominant else ctx.work()
        if tm1.get(slot, 0.0) > 0 and nl1 > 0:
            single_lane = {"step_s": min(ts), "avg_launch_ms": tm1[slot] / nl1 * 1e3, "launches_per_step": nl1,
                           "achieved": fl1 / tm1[slot] / 1e12, "frac": fl1 / tm1[slot] / 1e12 / F64_MATRIX_PEAK_TFLOPS,
                           "note": "the same step with DSMGP_OPT_LANES = 1, two untimed repetitions after one to rebuild the lists: "
                                   "flops of a launch over its own duration, as in the lines of rounds 1-4"}
        model.set_option(dsm.hipabi.OPT_LANES, args.lanes if args.lanes is not None else 0)
        ctx.set_profile(0 if args.no_profile else 1)
        roof["single_lane"] = single_lane

    if rank == 0:
        c = CONFIGS[args.config]
        out = {
            "metric": "fit!+predict wall-clock, DSMGP N=%dk D=%d" % (c["N"] // 1000, c["D"]),
            "value": per_step, "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per_step * 1e3, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"buildDSMGP K=4 splits V=3 sum children M={c['M']} N={c['N']} D={c['D']} "
                                   f"{'[IsoSE, IsoLinear]' if c.get('kvec') else 'IsoSE'} "
                                   f"({args.hyper} hyper-parameters: log l = {HYPER[args.hyper]['logl']:.4g}, log s = "
                                   f"{HYPER[args.hyper]['logs']:.4g}, logNoise = {HYPER[args.hyper]['lognoise']:.4g}) "
                                   f"depth {c['depth']}: {model.L} leaf GPs n={int(nobs.min())}..{int(nobs.max())}, "
                                   f"n_t={Xt.shape[0]} test rows x {int(ptr[-1] // Xt.shape[0])} leaves each; "
                                   f"fit! (Gram+Cholesky+forward solve+mll) + update! + predict",
                       "parallelism": f"leaves sharded over {world} GPU(s), all-gather of mll and of the aggregation's partial sums"
                                      + (f"; {n_sub} concurrent contexts per GPU" if n_sub > 1 else "")
                                      + (f"; {lanes} leaf lanes inside the context" if lanes > 1 else "")
                                      + (f"; exchange: {model.shard.exchange} over the {exchange_backend} process group"
                                         if world > 1 else "")},
            "matrix_tflops_fit_predict": matrix_flops_total / ((cats.get("total_fit", 0.0) + cats.get("total_predict", 0.0))
                                                               / args.steps) / 1e12 if world == 1 else None,
            "device_seconds_per_step": {k: v / args.steps for k, v in cats.items() if v > 0},
            "root_mll": root[0],
            "model_build_s": model.build_seconds,
        }
        out["step_s"] = spread(step_wall)                   # per-step wall seconds of the timed region (value = their mean + syncs)
        if not args.no_profile:
            key = step_fused if fused_dominant else step_upd
            out["dominant_launches_s_first5"] = [round(v, 6) for v in key[:5]]      # per step: device seconds during which the roofline
            out["dominant_launches_s_last5"] = [round(v, 6) for v in key[-5:]]      # kernel's launches ran (sustained-load droop)
        if step_clock:
            out["shader_clock_ghz_in_steps"] = {"first5": [round(v, 4) for v in step_clock[:min(5, args.steps)]],
                                                "last5": [round(v, 4) for v in step_clock[-min(5, args.steps):]],
                                                "sample_ms": clock_ms}
        if probe_before is not None:
            out["f64_mfma_probe"] = {"before_warmup": probe_before, "after_timed_loop": probe_after,
                                     "note": "register-only v_mfma_f64_16x16x4 loop, 8 waves per SIMD, whole chip; clock_ghz = "
                                             "shader clock held in the probe loop"}
        if world > 1:
            out["exchange_backend"] = exchange_backend      # "nccl" (= RCCL); "gloo" only if RCCL did not come up on this node
            out["ranks"] = ranks_info
            out["device_exchange_series"] = device_series   # None: no RCCL group, or the library's communicator did not come up
        if standalone is not None:
            out["drop_in_s"] = standalone["step_s"]
            out["standalone_fit_s"] = standalone["fit_s"]["median"]
            out["standalone_predict_s"] = standalone["predict_s"]["median"]
            out["standalone_predict_new_rows_s"] = standalone["predict_new_rows_s"]["median"]
            out["drop_in"] = standalone
        if roof is not None:
            out["roofline"] = roof
        if (world == 1 and args.config == "dsmgp_n100k_d8" and not args.simulate_shard and not args.no_other_configs
                and args.hyper == "survey"):
            # BASELINE configs 2, 3 and depth 4 on the same GPU, each a short series of its own (after every headline series)
            try:
                out["configs"] = other_configs(local_rank, torch)
            except Exception as e:      # noqa: BLE001 -- the headline line must be printed whatever happens to the extras
                out["configs"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(model, X, y, Xt, ptr, idx)
                out["speedup_vs_cpu_baseline"] = out["cpu_baseline"]["value"] / per_step
            except Exception as e:      # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(out))
    if td is not None:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
