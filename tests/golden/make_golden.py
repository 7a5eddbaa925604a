"""Generate the committed golden fixtures under tests/golden/.

The reference ships no tests or golden vectors and cannot be run here (no Julia),
so the vectors are produced by (a) closed forms, (b) 50-digit mpmath evaluations of the textbook GP
equations the reference implements (src/gaussianprocess.jl:82-137,163), and (c) the NumPy/LAPACK
oracle in oracle/, which (a) and (b) pin.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import gp as ogp, spn as ospn  # noqa: E402
import deepstructuredmixtures_amd as dsm  # noqa: E402
from deepstructuredmixtures_amd import tree as ptree  # noqa: E402
from deepstructuredmixtures_amd.datagen import uniform, normal  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 50


def mp_kernel(kind, loghyp, a, b):
    if kind == 0:
        l2 = mp.e ** (2 * mp.mpf(loghyp[0]))
        s2 = mp.e ** (2 * mp.mpf(loghyp[1]))
        z = sum((mp.mpf(float(x)) - mp.mpf(float(y))) ** 2 for x, y in zip(a, b))
        return s2 * mp.e ** (-z / (2 * l2))
    if kind == 1:
        s2 = mp.e ** (2 * mp.mpf(loghyp[-1]))
        tot = mp.mpf(0)
        for d in range(len(a)):
            l2 = mp.e ** (2 * mp.mpf(loghyp[d]))
            tot += mp.e ** (-((mp.mpf(float(a[d])) - mp.mpf(float(b[d]))) ** 2) / (2 * l2))
        return s2 * tot
    l2 = mp.e ** (2 * mp.mpf(loghyp[0]))
    return sum(mp.mpf(float(x)) * mp.mpf(float(y)) for x, y in zip(a, b)) / l2


def mp_gp(kind, loghyp, logNoise, X, y, mean, Xt):
    """50-digit alpha, mll, predictive mean and variance."""
    n = X.shape[0]
    noise = mp.e ** (2 * mp.mpf(logNoise))
    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            K[i, j] = mp_kernel(kind, loghyp, X[i], X[j])
        K[i, i] += noise + mp.mpf("1e-8")
    L = mp.cholesky(K)
    yc = mp.matrix([mp.mpf(float(v)) - mp.mpf(mean) for v in y])
    z = mp.lu_solve(L, yc)
    alpha = mp.lu_solve(L.T, z)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(n))
    mll = -(sum(yc[i] * alpha[i] for i in range(n)) + logdet + n * mp.log(2 * mp.pi)) / 2
    mus, vs = [], []
    for t in range(Xt.shape[0]):
        k = mp.matrix([mp_kernel(kind, loghyp, X[i], Xt[t]) for i in range(n)])
        mus.append(mp.mpf(mean) + sum(k[i] * alpha[i] for i in range(n)))
        v = mp.lu_solve(L, k)
        vs.append(mp_kernel(kind, loghyp, Xt[t], Xt[t]) - sum(v[i] ** 2 for i in range(n)) + noise)
    return ([float(a) for a in alpha], float(mll), [float(m) for m in mus], [float(v) for v in vs])


def gp_cases():
    cases = {}
    specs = [
        ("isose_n16_d2", 0, 16, 2, [np.log(0.4), np.log(1.3)], np.log(0.2)),
        ("isose_n40_d3", 0, 40, 3, [np.log(0.7), 0.1], -1.5),
        ("ardse_n24_d3", 1, 24, 3, [np.log(0.3), np.log(0.6), np.log(1.1), -0.2], np.log(0.3)),
        ("isolinear_n20_d2", 2, 20, 2, [np.log(0.8), 0.0], np.log(0.5)),
    ]
    for si, (name, kind, n, D, loghyp, logNoise) in enumerate(specs):
        X = uniform(100 + si, 0, n * D).reshape((n, D), order="F")
        y = np.sin(3.0 * X[:, 0]) + 0.1 * normal(200 + si, 0, n)
        Xt = uniform(300 + si, 0, 5 * D).reshape((5, D), order="F")
        mean = float(np.mean(y))
        k = ogp.make_kernel(kind, loghyp)
        for exact in (True, False):
            g = ogp.GaussianProcess(X, y, mean, k, logNoise, exact_dist=exact).update_cholesky()
            if exact:
                ge = g
        mu, var = ge.prediction(Xt)
        grad = ge.grad()
        alpha_mp, mll_mp, mu_mp, var_mp = mp_gp(kind, loghyp, logNoise, X, y, mean, Xt)
        # the LAPACK oracle must agree with the 50-digit evaluation before anything is stored
        assert np.allclose(ge.alpha, alpha_mp, rtol=1e-9, atol=1e-11), name
        assert abs(ge.mll() - mll_mp) < 1e-9 * max(1.0, abs(mll_mp)), name
        assert np.allclose(mu, mu_mp, rtol=1e-10, atol=1e-12), name
        assert np.allclose(var, var_mp, rtol=1e-9, atol=1e-12), name
        assert abs(g.mll() - mll_mp) < 1e-8 * max(1.0, abs(mll_mp)), name   # as-written distances too
        cases[name] = dict(kind=kind, X=X, y=y, Xt=Xt, mean=mean, loghyp=np.array(loghyp), logNoise=logNoise,
                           Knoisy=ge.noisy_kernel(), L=ge.L(), alpha=np.array(alpha_mp), mll=mll_mp,
                           mu=np.array(mu_mp), var=np.array(var_mp), grad=grad)
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f"{name}/{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "gp_small.npz"), **flat)
    return list(cases)


def gp_edge_cases():
    """n = 160 (one whole 128-block plus a partial one: the factorisation crosses a tile edge) for all three kernel kinds,
    with test points where the predictive variance cancels: three rows AT training inputs (rows on both sides of the
    tile edge), three 1e-7 away from them, three elsewhere.  Noise standard deviation 0.03: at a training point
    sigma^2 = k** + noise - |V|^2 loses three to four digits.  50-digit mpmath values; the oracle must match them first."""
    cases = {}
    specs = [
        ("isose_n160_d2", 0, 160, 2, [np.log(0.35), np.log(1.1)], np.log(0.03)),
        ("ardse_n160_d3", 1, 160, 3, [np.log(0.3), np.log(0.5), np.log(0.9), -0.1], np.log(0.03)),
        ("isolinear_n160_d3", 2, 160, 3, [np.log(0.9), 0.0], np.log(0.03)),
    ]
    for si, (name, kind, n, D, loghyp, logNoise) in enumerate(specs):
        X = uniform(500 + si, 0, n * D).reshape((n, D), order="F")
        y = np.sin(3.0 * X[:, 0]) * np.cos(2.0 * X[:, 1]) + 0.03 * normal(600 + si, 0, n)
        at = X[[5, 127, 128]]
        near = X[[40, 126, 159]] + 1e-7
        Xt = np.concatenate([at, near, uniform(700 + si, 0, 3 * D).reshape((3, D), order="F")])
        mean = float(np.mean(y))
        g = ogp.GaussianProcess(X, y, mean, ogp.make_kernel(kind, loghyp), logNoise, exact_dist=True).update_cholesky()
        mu, var = g.prediction(Xt)
        alpha_mp, mll_mp, mu_mp, var_mp = mp_gp(kind, loghyp, logNoise, X, y, mean, Xt)
        assert np.allclose(g.alpha, alpha_mp, rtol=1e-7, atol=1e-9), name          # conditioning ~ n s2 / noise ~ 2e5
        assert abs(g.mll() - mll_mp) < 1e-10 * max(1.0, abs(mll_mp)), name
        assert np.allclose(mu, mu_mp, rtol=1e-9, atol=1e-11), name
        assert np.allclose(var, var_mp, rtol=1e-8, atol=1e-12), name
        cases[name] = dict(kind=kind, X=X, y=y, Xt=Xt, mean=mean, loghyp=np.array(loghyp), logNoise=logNoise,
                           alpha=np.array(alpha_mp), mll=mll_mp, mu=np.array(mu_mp), var=np.array(var_mp))
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f"{name}/{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "gp_edge.npz"), **flat)
    return {n: (float(np.min(c["var"])), float(c["mll"])) for n, c in cases.items()}


def analytic():
    """n=1 and n=2 closed forms (IsoSE), written out by hand."""
    out = {}
    # n = 1: K = s2 + noise + eps; alpha = y/K; mll = -(y^2/K + log K + log 2pi)/2
    logl, logs, logn = 0.3, -0.2, -0.7
    s2, noise = np.exp(2 * logs), np.exp(2 * logn)
    x, yv, xt = 0.25, 0.8, 0.6
    Kd = s2 + noise + 1e-8
    kxt = s2 * np.exp(-0.5 * (x - xt) ** 2 / np.exp(2 * logl))
    out["n1"] = dict(logl=logl, logs=logs, logNoise=logn, x=x, y=yv, xt=xt, mean=0.0, alpha=yv / Kd,
                     mll=-(yv * yv / Kd + np.log(Kd) + np.log(2 * np.pi)) / 2, mu=kxt * yv / Kd,
                     var=s2 - kxt * kxt / Kd + noise)
    # n = 2
    x1, x2, y1, y2 = 0.1, 0.9, 0.5, -0.3
    a = s2 + noise + 1e-8
    b = s2 * np.exp(-0.5 * (x1 - x2) ** 2 / np.exp(2 * logl))
    det = a * a - b * b
    al = np.array([(a * y1 - b * y2) / det, (a * y2 - b * y1) / det])
    k1 = s2 * np.exp(-0.5 * (x1 - xt) ** 2 / np.exp(2 * logl))
    k2 = s2 * np.exp(-0.5 * (x2 - xt) ** 2 / np.exp(2 * logl))
    quad = (a * k1 * k1 - 2 * b * k1 * k2 + a * k2 * k2) / det
    out["n2"] = dict(logl=logl, logs=logs, logNoise=logn, x=[x1, x2], y=[y1, y2], xt=xt, mean=0.0,
                     alpha=al.tolist(), mll=-((y1 * al[0] + y2 * al[1]) + np.log(det) + 2 * np.log(2 * np.pi)) / 2,
                     mu=k1 * al[0] + k2 * al[1], var=s2 - quad + noise)
    with open(os.path.join(OUT, "analytic.json"), "w") as f:
        json.dump(out, f, indent=1)


def config1():
    """BASELINE config 1: README 1-D sinusoid N=100, IsoSE(1,1), K=4 splits, V=3 sum children, M=10
    (README.md:35-51), with a fixed noise vector instead of randn."""
    N = 100
    x = np.linspace(0.0, 1.0, N)
    y = np.sin(x * 4 * np.pi + normal(42, 0, N) * 0.2)
    xt = np.linspace(0.5, 1.5, 100).reshape(-1, 1)[:50] * 0.98   # stay inside the data range too
    xt = np.concatenate([xt[:25], np.linspace(0.05, 0.95, 25).reshape(-1, 1)])
    model = dsm.buildDSMGP(x.reshape(-1, 1), y, 3, 4, M=10, kernel=dsm.IsoSE(1.0, 1.0),
                           meanFun=dsm.ConstMean(float(np.mean(x))), seed=11, fit_now=False)
    X = x.reshape(-1, 1)
    gps = ospn.make_leaf_gps(model.root, X, y, exact_dist=True)
    D = ospn.get_overlap(model.root, model.L)
    census = ospn.fit(model.root, gps, D, tau=0.05)
    leaf_mll = np.array([g.mll() for g in gps])
    z = ospn.update(model.root, gps)
    mu, var = ospn.predict(model.root, gps, xt)
    ptr = np.concatenate([[0], np.cumsum([lf.nobs for lf in model.leaves])])
    idx = np.concatenate([lf.obs for lf in model.leaves])
    np.savez_compressed(os.path.join(OUT, "config1.npz"), x=x, y=y, xt=xt, obs_ptr=ptr, obs_idx=idx,
                        leaf_mll=leaf_mll, root_mll=z, mu=mu, var=var,
                        census=np.array([census[k] for k in ("full", "copy", "prefix", "lowrank_as_full", "leading_as_full")]))
    print("config1: L =", model.L, "census", census, "root mll", z)


def tree_small():
    """A D=2 DSMGP small enough for the oracle, used by GPU parity and host-aggregation tests."""
    N, D = 600, 2
    X = uniform(7, 0, N * D).reshape((N, D), order="F")
    y = np.sin(6 * X[:, 0]) * np.cos(4 * X[:, 1]) + 0.1 * normal(8, 0, N)
    Xt = uniform(9, 0, 80 * D).reshape((80, D), order="F")
    model = dsm.buildDSMGP(X, y, 2, 4, M=20, kernel=dsm.IsoSE(np.log(0.3), 0.0), logNoise=np.log(0.1),
                           seed=3, fit_now=False)
    gps = ospn.make_leaf_gps(model.root, X, y, exact_dist=True)
    Dm = ospn.get_overlap(model.root, model.L)
    ospn.fit(model.root, gps, Dm)
    leaf_mll = np.array([g.mll() for g in gps])
    z = ospn.update(model.root, gps)
    mu, var = ospn.predict(model.root, gps, Xt)
    np.savez_compressed(os.path.join(OUT, "tree_small.npz"), X=X, y=y, Xt=Xt, leaf_mll=leaf_mll, root_mll=z,
                        mu=mu, var=var, leaf_sizes=np.array([lf.nobs for lf in model.leaves]))
    print("tree_small: L =", model.L, "root mll", z)


if __name__ == "__main__":
    print("gp cases:", gp_cases())
    print("tile-edge cases (min var, mll):", gp_edge_cases())
    analytic()
    config1()
    tree_small()
