"""TEST INFRASTRUCTURE -- CPU restatement of the reference's random tree builder (parity unpinned, see oracle/__init__.py).

Literal restatement of `/root/reference/src/treeStructure.jl:4-307`: `buildTree` (:4-21), `getSplits` (:23-129), `_buildSplit`
(:131-210), `_buildSum` (:212-243), `_buildGP` (:245-307), statement for statement -- every recursion receives the region's own
rows (the reference passes `view(X, idx, :)`), carries whole `lowerBound` / `upperBound` VECTORS and recomputes its index sets from
them with `findall`, exactly as the reference does; nothing is shared with the product's builders (`deepstructuredmixtures_amd/
tree.py`, `csrc/host_tree.cpp`), which work on one column and two scalars.  Only `tests/` may import this module: it is the second
witness of the native builder, never a code path of the product.

Random draws: Julia's RNG stream cannot be reproduced (no Julia here, Distributions' samplers are unpinned), so the reference's
four draw sites -- `rand(Beta(2, 2))` (:52), `rand(1:2)` (:66), `rand(Categorical(phi))` (:236), `rand(Dirichlet(k, 1.0))`
(:260) -- take their uniforms from the portable counter stream of SURVEY 8(d) (SplitMix64 in counter mode), restated here in
plain Python integers: Beta(2, 2) as the median of three uniforms, `rand(1:2)` as 1 + floor(2u), the categorical by inverse
cdf, Dirichlet(1) as normalised exponentials.  Same distributions as the reference's, same stream positions as the product's.

Indices are 0-based; `observations` are ascending original row indices (:15, :181 `observations[idx]`).
"""
import math

import numpy as np

_MASK = (1 << 64) - 1


class CounterStream:
    """SplitMix64 in counter mode: draw i (0-based) of stream `seed` is mix(seed + (i + 1) * GAMMA); the uniform is its top 53
    bits / 2^53.  Plain integers -- a restatement independent of the vectorised generator of the product."""

    def __init__(self, seed):
        self.seed = int(seed) & _MASK
        self.pos = 0

    def uniform(self):
        self.pos += 1
        z = (self.seed + self.pos * 0x9E3779B97F4A7C15) & _MASK
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        z ^= z >> 31
        return (z >> 11) * (1.0 / 9007199254740992.0)

    def beta22(self):                   # rand(Beta(2, 2)): the second order statistic of three uniforms
        return sorted([self.uniform(), self.uniform(), self.uniform()])[1]

    def one_or_two(self):               # rand(1:2)
        return 1 + int(self.uniform() * 2)

    def categorical(self, phi):         # rand(Categorical(phi)), 0-based: first index whose cumulative weight exceeds u * total
        c = np.cumsum(np.asarray(phi, dtype=np.float64))
        t = self.uniform() * c[-1]
        for k in range(len(c)):
            if t < c[k]:
                return k
        return len(c) - 1

    def dirichlet1(self, k):            # rand(Dirichlet(k, 1.0))
        e = -np.log(1.0 - np.array([self.uniform() for _ in range(k)]))
        return e / e.sum()


def _median(v):
    """Julia's `median`: middle element, or `middle(a, b) = (a + b) / 2` of the two middle ones (Statistics.jl)."""
    s = np.sort(np.asarray(v, dtype=np.float64))
    n = s.size
    return float(s[n // 2]) if n % 2 else float((s[n // 2 - 1] + s[n // 2]) / 2)


def get_splits(X, lowerBound, upperBound, minData, eps, K, d, rng, depth=1):
    """`getSplits` (`src/treeStructure.jl:23-129`)."""
    K_ = depth ** 2                                                               # :33
    s = []
    l = max(lowerBound[d], float(np.min(X[:, d])))                                # :36
    u = min(upperBound[d], float(np.max(X[:, d])))                                # :37
    v = u - l
    idx = np.flatnonzero((X[:, d] > l) & (X[:, d] <= u))                          # :40
    if len(idx) > minData * 2:                                                    # :41
        z1 = z2 = 0
        c = 0
        m = _median(X[idx, d])                                                    # :49 (the mean of :48 is overwritten)
        while z1 == 0 or z2 == 0:                                                 # :51
            a = rng.beta22() * v + l                                              # :52
            s_new = float(eps * a + (1 - eps) * m)                                # :54
            z1 = int(np.sum(X[idx, d] <= s_new))                                  # :56
            z2 = int(np.sum(X[idx, d] > s_new))                                   # :57
            c += 1
            if c > 100:                                                           # :61-64
                return s
        zi = rng.one_or_two()                                                     # :66
        if zi == 1:
            if z1 > minData and K_ < K:                                           # :68
                ub = np.array(upperBound, dtype=np.float64)
                ub[d] = s_new
                s += get_splits(X, lowerBound, ub, minData, eps, K, d, rng, depth + 1)
                K_ += 1                                                           # :81
            if z2 > minData and K_ < K:                                           # :83
                lb = np.array(upperBound, dtype=np.float64)                       # :84 `lb = copy(upperBound)` as written: only
                lb[d] = s_new                                                     #     index d is read below (:36), so harmless
                s += get_splits(X, lb, upperBound, minData, eps, K, d, rng, depth + 1)
        else:
            if z2 > minData and K_ < K:                                           # :97
                lb = np.array(upperBound, dtype=np.float64)                       # :98 (same)
                lb[d] = s_new
                s += get_splits(X, lb, upperBound, minData, eps, K, d, rng, depth + 1)
                K_ += 1                                                           # :110
            if z1 > minData and K_ < K:                                           # :112
                ub = np.array(upperBound, dtype=np.float64)
                ub[d] = s_new
                s += get_splits(X, lowerBound, ub, minData, eps, K, d, rng, depth + 1)
        s.append(s_new)                                                           # :126
    return s


def _build_gp(X, y, lowerBound, upperBound, cfg, observations, rng):
    """`_buildGP` (`src/treeStructure.jl:245-307`): a region = one GP, or a sum node over one GP per kernel with log Dirichlet(1)
    weights (:258-286).  mean = ConstMean(mean(y)) when the config has none (:271, :292)."""
    node = dict(kind="region", lb=lowerBound, ub=upperBound, obs=np.asarray(observations, dtype=np.int64),
                mean=(float(np.mean(y)) if len(y) else 0.0) if cfg["meanFun"] is None else cfg["meanFun"], weights=None)
    if cfg["n_kernels"] > 0:                                                      # config.kernels isa Vector
        node["weights"] = rng.dirichlet1(cfg["n_kernels"])                        # :260
    return node


def _build_split(X, y, lowerBound, upperBound, cfg, depth, observations, rng, d=0):
    """`_buildSplit` (`src/treeStructure.jl:131-210`); membership of a child region is (lb, ub] on dimension d (:181)."""
    s = sorted(get_splits(X, lowerBound, upperBound, cfg["minData"], cfg["bnoise"], cfg["K"], d, rng))      # :150-157
    split = [(d, si) for si in s] + [(d, float(upperBound[d]))]                   # :159-165
    node = dict(kind="split", lb=lowerBound, ub=upperBound, dim=d, thr=[si for _, si in split], children=[])
    lb = np.array(lowerBound, dtype=np.float64)                                   # :171
    ub = np.array(upperBound, dtype=np.float64)
    if s:                                                                         # :174
        for _, si in split:
            lb_ = lb.copy()
            ub_ = ub.copy()
            ub_[d] = si
            idx = np.flatnonzero((X[:, d] > lb_[d]) & (X[:, d] <= ub_[d]))        # :181
            if depth < cfg["depth"] and len(idx) > cfg["minData"]:                # :182
                if cfg["sumRoot"]:
                    child = _build_sum(X[idx], y[idx], lb_, ub_, cfg, depth, observations[idx], rng)
                else:                                                             # :190: `d` not forwarded (dimension 1 again),
                    child = _build_split(X[idx], y[idx], lb_, ub_, cfg, depth, observations[idx], rng)    # depth not increased
            else:
                child = _build_gp(X[idx], y[idx], lb_, ub_, cfg, observations[idx], rng)
            node["children"].append(child)
            lb[d] = si                                                            # :198
        return node
    l, u = lowerBound[d], upperBound[d]                                           # :202-203
    idx = np.flatnonzero((X[:, d] > l) & (X[:, d] <= u))
    return _build_gp(X[idx], y[idx], np.array(lowerBound, dtype=np.float64), np.array(upperBound, dtype=np.float64), cfg,
                     observations[idx], rng)


def _build_sum(X, y, lowerBound, upperBound, cfg, depth, observations, rng):
    """`_buildSum` (`src/treeStructure.jl:212-243`): V children, each a split on a dimension drawn in proportion to the data
    range of the region; initial weights -log V (:226)."""
    V = cfg["V"]
    node = dict(kind="sum", lb=lowerBound, ub=upperBound, children=[], logweights=np.full(V, -math.log(V)))
    phi = np.array([float(np.max(X[:, j]) - np.min(X[:, j])) for j in range(X.shape[1])])                   # :232-233
    if not phi.sum() > 0:
        raise ValueError("region without extent: the reference draws from Categorical(NaN) here and throws")
    phi = phi / phi.sum()                                                         # :234
    for _ in range(V):
        d = rng.categorical(phi)                                                  # :236
        node["children"].append(_build_split(X, y, lowerBound, upperBound, cfg, depth + 1, observations, rng, d=d))
    return node


def build_tree(X, y, minData, K, V, depth, bnoise, sumRoot, n_kernels=0, meanFun=None, seed=7):
    """`buildTree` (`src/treeStructure.jl:4-21`) with the config fields of `build` (:405-437): K = cuts per split node
    (`getSplits` stops at K_ >= K), V = children per sum node, n_kernels > 0 = `config.kernels isa Vector` of that length."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    N, D = X.shape
    assert N == len(y) and np.all(np.isfinite(X))                                 # :7, :15
    cfg = dict(minData=int(minData), K=int(K), V=int(V), depth=int(depth), bnoise=float(bnoise), sumRoot=bool(sumRoot),
               n_kernels=int(n_kernels), meanFun=meanFun)
    rng = CounterStream(seed)
    lowerBound = np.full(D, -np.inf)
    upperBound = np.full(D, np.inf)
    observations = np.arange(N, dtype=np.int64)
    if sumRoot:
        return _build_sum(X, y, lowerBound, upperBound, cfg, 0, observations, rng)
    return _build_split(X, y, lowerBound, upperBound, cfg, 0, observations, rng)


def table(root):
    """The tree as flat arrays in creation (pre-)order, the layout `dsmgp_tree_export` writes: kind (0 region, 1 split, 2 sum),
    parent, split_dim (-1 unless split), lb / ub (n x D), thr CSR, obs CSR, per region its mean and kernel weights."""
    kind, parent, sdim, lb, ub, thr_ptr, thr, obs_ptr, obs, mean, weights = [], [], [], [], [], [0], [], [0], [], [], []
    code = {"region": 0, "split": 1, "sum": 2}

    def walk(node, par):
        i = len(kind)
        kind.append(code[node["kind"]])
        parent.append(par)
        sdim.append(node["dim"] if node["kind"] == "split" else -1)
        lb.append(np.array(node["lb"], dtype=np.float64))
        ub.append(np.array(node["ub"], dtype=np.float64))
        if node["kind"] == "split":
            thr.extend(node["thr"])
        if node["kind"] == "region":
            obs.extend(node["obs"].tolist())
            mean.append(node["mean"])
            weights.append(node["weights"])
        thr_ptr.append(len(thr))
        obs_ptr.append(len(obs))
        for c in node.get("children", []):
            walk(c, i)

    walk(root, -1)
    return dict(kind=np.array(kind, dtype=np.int32), parent=np.array(parent, dtype=np.int32), split_dim=np.array(sdim, dtype=np.int32),
                lb=np.array(lb), ub=np.array(ub), thr_ptr=np.array(thr_ptr, dtype=np.int64), thr=np.array(thr, dtype=np.float64),
                obs_ptr=np.array(obs_ptr, dtype=np.int64), obs=np.array(obs, dtype=np.int64), mean=mean, weights=weights)
