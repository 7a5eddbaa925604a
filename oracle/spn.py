"""ORACLE -- test infrastructure, not product code (see oracle/gp.py for the rules and the
"parity unpinned" statement).

CPU restatement of the tree-level callers of the per-leaf GP arithmetic: the shared-Cholesky
scheduler `fit!` (src/fit.jl:71-292), `fit_naive!` (:294-304), the predict recursions
(src/common.jl:101-313), `update!`/`infer!`/`mll` (src/common.jl:323-355, src/optimize.jl:18-39) and
the gradient back-propagation `∇mll!` (src/optimize.jl:42-89).  The recursions are written as the
reference writes them (two prediction passes, log-domain sums); trees are duck-typed: a node has
`.kind` in {"gp","split","sum"}, `.children`, sums have `.logweights` and `.of_gps`, splits have
`.split` = [(dim, threshold)], leaves have `.obs`, `.kernelid`, `.leaf`, `.mean.m`, `.kernel`
(object with `.kind` and `.loghyp()`), `.logNoise`.
"""
import numpy as np
import scipy.linalg as sla

from . import gp as ogp

EPS = ogp.EPS


def get_leaves(node):
    """src/fit.jl:9-10"""
    if node.kind == "gp":
        return [node]
    out = []
    for c in node.children:
        out.extend(get_leaves(c))
    return out


def make_leaf_gps(root, X, y, exact_dist=False):
    """One oracle GaussianProcess per leaf, as _buildGP constructs them (src/treeStructure.jl:288-305)."""
    gps = []
    for lf in get_leaves(root):
        hyp = lf.kernel.loghyp()
        k = ogp.make_kernel(lf.kernel.kind, hyp)
        gps.append(ogp.GaussianProcess(X[lf.obs], y[lf.obs], lf.mean.m, k, lf.logNoise, exact_dist))
    return gps


def set_hyper(root, gps):
    """Copy the current hyper-parameters of the tree's leaves into the oracle GPs (setparams!)."""
    for lf, g in zip(get_leaves(root), gps):
        g.kernel = ogp.make_kernel(lf.kernel.kind, lf.kernel.loghyp())
        g.logNoise = lf.logNoise


def get_overlap(root, L):
    """src/fit.jl:12-39 with the observation BitArrays replaced by index sets."""
    D = np.zeros((L, L))

    def rec(node):
        if node.kind == "gp":
            return [node]
        if node.kind == "split":
            return [x for c in node.children for x in rec(c)]
        r = [rec(c) for c in node.children]
        for i in range(len(r)):
            for j in range(i + 1, len(r)):
                for nn in r[i]:
                    sn = set(nn.obs.tolist())
                    for mm in r[j]:
                        sm = set(mm.obs.tolist())
                        same = 1 if nn.kernelid == mm.kernelid else 0
                        dn = len(sn - sm) * same      # Δn :28
                        dm = len(sm - sn) * same      # Δm :29
                        D[nn.leaf, mm.leaf] = 1.0 - dn / len(sn)
                        D[mm.leaf, nn.leaf] = 1.0 - dm / len(sm)
        return [x for rr in r for x in rr]

    rec(root)
    return D


def fit(root, gps, D, tau=0.05, as_written=False, census_only=False):
    """fit!(spn, D, gpmap; τ) (src/fit.jl:71-122) with fitcontained! (:124-292).

    as_written=False ("lean"): one potrf per leaf, the low-rank row-deletion branches replaced by
    a full factorisation (they are numerically defective).  as_written=True additionally
    reproduces the unconditional extra update_cholesky!(jGP) of :105 for timing.
    Returns a census dict of the branch the REFERENCE takes per leaf: counts under "full", "copy", "prefix",
    "lowrank_as_full" (its row-deletion branch with rows to delete: :174-201, computed in full here) and
    "leading_as_full" (the same branch with nothing to delete: the leaf's list is a leading part of its main leaf's
    and the reference takes the leading block of that factor -- exact there, computed in full here), plus
    "lowrank_leaves" / "leading_leaves", the leaf ids of the last two.
    census_only=True walks the decisions without any arithmetic (gps may be None; a prefix continuation is assumed
    to succeed): the census of a model too large to factorise on the host."""
    if census_only:
        gps = [_NoGP() for _ in get_leaves(root)]
    leaves = get_leaves(root)
    n = len(leaves)
    processed = np.zeros(n, dtype=bool)
    counts = np.zeros(n, dtype=np.int64)
    S = np.zeros(n, dtype=np.int64)
    for j in range(n):
        i = int(np.argmax(D[:, j] * D[j, :]))                    # :79
        counts[i] += 1
        S[j] = i
    order = sorted(range(n), key=lambda j: counts[j])            # sort! :86 (stable)
    census = {"full": 0, "copy": 0, "prefix": 0, "lowrank_as_full": 0, "leading_as_full": 0,
              "lowrank_leaves": [], "leading_leaves": []}
    for j in order:
        if processed[j]:
            continue
        i = int(S[j])
        mainGP = gps[i]
        if not processed[i]:
            mainGP.update_cholesky()                             # :98
            processed[i] = True
            census["full"] += 1
            if i == j:
                if as_written:
                    gps[j].update_cholesky()                     # :105 again on the same leaf
                continue
        jGP = gps[j]
        processed[j] = True
        if as_written:
            jGP.update_cholesky()                                # :105 (F3)
        lj, li = leaves[j], leaves[i]
        if i == j or li.kernelid != lj.kernelid or lj.obs[0] < li.obs[0]:
            jGP.update_cholesky()                                # :107-112
            census["full"] += 1
            continue
        ione = D[i, j] == 1.0
        jone = D[j, i] == 1.0
        if ione and jone:
            if not census_only:
                jGP.factors = mainGP.factors.copy()              # :141-142
                jGP.alpha = mainGP.alpha.copy()
                jGP.info = mainGP.info
            census["copy"] += 1
        elif ione and not jone:
            _fit_superset(jGP, lj, mainGP, li, tau, census)      # :208-292
        elif jone and not ione:
            # :145-206.  e = position of max(j.obs) in the main leaf's list; toupdate = main.obs[1:e] \ j.obs (:170);
            # j.obs is a subset of main.obs[1:e], so |toupdate| = e - |j.obs|
            e = int(np.searchsorted(li.obs, lj.obs[-1])) + 1
            ndel = e - lj.nobs
            jGP.update_cholesky()                                # every arm of the branch -> full here
            if ndel / lj.nobs < tau:                             # :173
                if ndel > 0:                                     # row deletions (:179-187): defective (F4)
                    census["lowrank_as_full"] += 1
                    census["lowrank_leaves"].append(int(j))
                else:                                            # nothing to delete: leading block of the main factor
                    census["leading_as_full"] += 1
                    census["leading_leaves"].append(int(j))
            else:
                census["full"] += 1                              # :203-205
        else:
            jGP.update_cholesky()                                # :124-130
            census["full"] += 1
    census["lowrank_leaves"].sort()
    census["leading_leaves"].sort()
    return census


class _NoGP:
    """Stand-in leaf for fit(census_only=True): every numerical step is a no-op."""
    factors = alpha = None
    info = 0

    def update_cholesky(self):
        return self


def _fit_superset(jGP, lj, mainGP, li, tau, census):
    """fitcontained!(..., Val(true), Val(false), τ): src/fit.jl:208-292."""
    maxM = li.obs[-1]
    minJ, minM = lj.obs[0], li.obs[0]
    pos = np.flatnonzero(lj.obs == maxM)
    s1 = lj.obs[: pos[0] + 1] if pos.size else lj.obs[:0]        # :247
    s2 = li.obs                                                  # :248
    toupdate = np.setdiff1d(li.obs, s1)                          # :249
    if len(s1) != len(s2) and minJ == minM:                      # :251
        jGP.update_cholesky()
        census["full"] += 1
        return
    if (len(toupdate) / lj.nobs) < tau:                          # :256
        if len(toupdate) > 0:                                    # (cannot happen: D[main, j] = 1 puts main.obs inside s1)
            jGP.update_cholesky()                                # low-rank deletes -> full (F4)
            census["lowrank_as_full"] += 1
            census["lowrank_leaves"].append(int(lj.leaf))
            return
        if isinstance(jGP, _NoGP):
            census["prefix"] += 1
            return
        F = jGP.noisy_kernel()                                   # :218-230
        p = len(s1)
        F[:p, :p] = mainGP.factors[:p, :p]                       # :276
        F, info = ogp.chol_continue(F, p + 1)                    # :278
        if info == 0 and np.all(np.diag(F) >= 0.0):              # :280-283
            jGP.factors = F
            jGP.info = 0
            jGP.solve_alpha()
            census["prefix"] += 1
        else:
            jGP.update_cholesky()                                # :285
            census["full"] += 1
    else:
        jGP.update_cholesky()                                    # :289
        census["full"] += 1


def fit_naive(root, gps):
    """src/fit.jl:294-304"""
    for g in gps:
        g.update_cholesky()


# ------------------------------------------------------------------------------- mll / weights

def lse_rows(x):
    """lse(x; dims=2): src/common.jl:309-313"""
    m = np.max(x, axis=1, keepdims=True)
    return (np.log(np.sum(np.exp(x - m), axis=1, keepdims=True)) + m)[:, 0]


def logsumexp(v):
    v = np.asarray(v, dtype=np.float64)
    m = np.max(v)
    return float(np.log(np.sum(np.exp(v - m))) + m)


def mll(node, gps):
    """src/optimize.jl:18-25"""
    if node.kind == "gp":
        return gps[node.leaf].mll()
    if node.kind == "split":
        return sum(mll(c, gps) for c in node.children)
    K = len(node.children)
    return logsumexp([-np.log(K) + mll(c, gps) for c in node.children])


def mll_table(node, gps, tab):
    """mll!(node, ℓ): src/optimize.jl:27-39"""
    if node.kind == "gp":
        v = gps[node.leaf].mll()
    elif node.kind == "split":
        v = sum(mll_table(c, gps, tab) for c in node.children)
    else:
        K = len(node.children)
        v = logsumexp([-np.log(K) + mll_table(c, gps, tab) for c in node.children])
    tab[node.id] = v
    return v


def update(node, gps):
    """update!(node): src/common.jl:323-334 -- returns z, sets normalised logweights in place."""
    if node.kind == "gp":
        return gps[node.leaf].mll()
    if node.kind == "split":
        return sum(update(c, gps) for c in node.children)
    K = len(node.children)
    lw = np.array([-np.log(K) + update(c, gps) for c in node.children])
    z = logsumexp(lw)
    node.logweights = lw - z
    return z


def infer(node, gps):
    """infer!(node): src/common.jl:336-355"""
    if node.kind == "gp":
        return gps[node.leaf].mll()
    if node.kind == "split":
        return sum(infer(c, gps) for c in node.children)
    K = len(node.children)
    lw = np.array([-np.log(K) + infer(c, gps) for c in node.children])
    z = logsumexp(lw)
    node.logweights = (lw - z) if node.of_gps else np.full(K, -np.log(K))
    return z


# ------------------------------------------------------------------------------- predict

def getchild(node, x):
    """src/common.jl:101-122, 0-based child index."""
    idx = np.full(x.shape[0], -1, dtype=np.int64)
    for n in range(x.shape[0]):
        k = 0
        while idx[n] < 0:
            d, s = node.split[k]
            if k == 0:
                accept = x[n, d] <= s
            else:
                accept = (x[n, d] <= s) and (x[n, d] > node.split[k - 1][1])
            if accept:
                idx[n] = k
            k += 1
    return idx


def _leaf_prediction(node, gps, x):
    return gps[node.leaf].prediction(x)


def _minpredict(node, gps, x):
    """src/common.jl:151-173"""
    if node.kind == "gp":
        return _leaf_prediction(node, gps, x)[0]
    if node.kind == "split":
        idx = getchild(node, x)
        mu = np.zeros(x.shape[0])
        for k, c in enumerate(node.children):
            j = np.flatnonzero(idx == k)
            mu[j] = _minpredict(c, gps, x[j])
        return mu
    mu = np.full(x.shape[0], np.inf)
    for c in node.children:
        mu = np.minimum(mu, _minpredict(c, gps, x))
    return mu


def _predict(node, gps, x, mumin):
    """src/common.jl:134-143 (leaf), :181-196 (split), :275-292 (sum)"""
    if node.kind == "gp":
        mu, s2 = _leaf_prediction(node, gps, x)
        s2 = s2.copy()
        s2[s2 <= 0] = EPS                                        # :137
        assert np.all(mu >= mumin)                               # :138
        with np.errstate(divide="ignore"):
            return np.log(mu - mumin), np.log(mu ** 2), np.log(s2)
    if node.kind == "split":
        idx = getchild(node, x)
        lm = np.zeros(x.shape[0])
        lm2 = np.zeros(x.shape[0])
        ls = np.zeros(x.shape[0])
        for k, c in enumerate(node.children):
            j = np.flatnonzero(idx == k)
            lm[j], lm2[j], ls[j] = _predict(c, gps, x[j], mumin[j])
        return lm, lm2, ls
    K = len(node.children)
    lm = np.zeros((x.shape[0], K))
    lm2 = np.zeros((x.shape[0], K))
    ls = np.zeros((x.shape[0], K))
    for k, c in enumerate(node.children):
        a, b, d = _predict(c, gps, x, mumin)
        lm[:, k] = a + node.logweights[k]
        lm2[:, k] = b + node.logweights[k]
        ls[:, k] = d + node.logweights[k]
    return lse_rows(lm), lse_rows(lm2), lse_rows(ls)


def predict(node, gps, x):
    """predict(node, x): src/common.jl:175-179 (leaf), :243-254 (split), :294-302 (sum)"""
    x = np.asarray(x, dtype=np.float64)
    if node.kind == "gp":
        mumin = _minpredict(node, gps, x)
        lm, _, ls = _predict(node, gps, x, mumin - 1)
        return np.exp(lm) + mumin - 1, np.exp(ls)
    if node.kind == "split":
        idx = getchild(node, x)
        mu = np.zeros(x.shape[0])
        s2 = np.zeros(x.shape[0])
        for k, c in enumerate(node.children):
            j = np.flatnonzero(idx == k)
            mu[j], s2[j] = predict(c, gps, x[j])
        return mu, s2
    mumin = _minpredict(node, gps, x)
    lm, lm2, ls = _predict(node, gps, x, mumin - 1)
    mu = np.exp(lm) + mumin - 1                                  # :299
    v = np.exp(ls) + (np.exp(lm2) - mu ** 2)                     # :300
    return mu, v


def _predict_poe(node, gps, x):
    """src/common.jl:145-149 (leaf), :198-208 (split)"""
    if node.kind == "gp":
        mu, s2 = _leaf_prediction(node, gps, x)
        return mu, 1.0 / s2
    mu = np.zeros(x.shape[0])
    t = np.zeros(x.shape[0])
    for c in node.children:
        m_, t_ = _predict_poe(c, gps, x)
        t += t_
        mu += t_ * m_
    return mu / t, t


def predict_poe(root, gps, x):
    """src/common.jl:256-260"""
    mu, t = _predict_poe(root, gps, np.asarray(x, dtype=np.float64))
    return mu, 1.0 / t


def predict_gpoe(root, gps, x):
    """src/common.jl:211-222,263-267 (β = 1/#children of the root)"""
    x = np.asarray(x, dtype=np.float64)
    mu = np.zeros(x.shape[0])
    t = np.zeros(x.shape[0])
    beta = 1.0 / len(root.children)
    for c in root.children:
        m_, t_ = _predict_poe(c, gps, x)
        t += beta * t_
        mu += beta * t_ * m_
    return mu / t, 1.0 / t


def predict_rbcm(root, gps, x):
    """src/common.jl:224-241,269-273"""
    x = np.asarray(x, dtype=np.float64)
    g0 = gps[get_leaves(root)[0].leaf]                           # leftGP :227
    s = ogp.prior_diag(g0.kernel, x) + g0.getnoise()             # :228
    C = 1.0 / s                                                  # :230
    mu = np.zeros(x.shape[0])
    for c in root.children:
        m_, t_ = _predict_poe(c, gps, x)
        s_ = 1.0 / t_
        beta = 0.5 * (np.log(s) - np.log(s_))                    # :235
        C = C + (beta * t_) - (beta / s)                         # :236
        mu = mu + m_ * (beta * t_)                               # :237
    return mu / C, 1.0 / C


# ------------------------------------------------------------------------------- gradients

def grad_tree(root, gps, n_hyp, leaf_weights=None):
    """updategradients!(spn) + ∇mll!(spn, 0, 0, L, L[root], grad): src/fit.jl:306-311,
    src/optimize.jl:42-89.  One shared hyper-vector (per kernel id for sums over GPs).
    leaf_weights (a row of the overlap matrix, indexed by leaf): the finetune! methods, src/optimize.jl:91-150,
    which multiply every leaf's term by D[gpmap.x[node.id]] (:101)."""
    tab = {}
    mll_table(root, gps, tab)
    logS = tab[root.id]
    grad = np.zeros(n_hyp)

    def rec(node, dparent, lrho, g):
        if node.kind == "gp":
            w = np.exp(-logS + lrho + tab[node.id] + dparent)    # :48
            if leaf_weights is not None:
                w = w * leaf_weights[node.leaf]                  # :101
            g += gps[node.leaf].grad() * w                       # :49
        elif node.kind == "split":
            for c in node.children:
                lp = tab[node.id] - tab[c.id]                    # :59
                rec(c, dparent + lp, lrho, g)
        elif node.of_gps:
            c0 = 0
            for c in node.children:                              # :76-89
                nn = gps[c.leaf].kernel.nl() + 2
                rec(c, dparent, lrho, g[c0:c0 + nn])
                c0 += nn
        else:
            K = len(node.children)
            for c in node.children:
                rec(c, -np.log(K) + dparent, np.log(K) + lrho, g)  # :72

    rec(root, 0.0, 0.0, grad)
    return grad


def finetune(root, gps, D, step, iterations, lam=0.5, tau=0.05):
    """finetune!(spn, D, gpmap, optim; iterations, λ): src/finetuning.jl:8-87, restated with the oracle's pieces.
    `step(hyp, grad)` is the optimiser's increment (Flux.Optimise.apply!, :55), added to the leaf's vector (:56).
    Returns (per-leaf hyper-vectors, history).  One kernel id."""
    leaves = get_leaves(root)
    hyp = [np.concatenate([lf.kernel.loghyp(), [lf.logNoise]]) for lf in leaves]       # :22
    n_hyp = hyp[0].size
    hist, c = [], 0
    for it in range(1, iterations + 1):
        ell = 0.0
        for j, lf in enumerate(leaves):
            for l2, g in zip(leaves, gps):                                             # setparams!(spn, hyp_) :41
                g.kernel = ogp.make_kernel(l2.kernel.kind, hyp[j][:-1])
                g.logNoise = float(hyp[j][-1])
            fit(root, gps, D, tau)                                                     # :44
            tab = {}
            mll_table(root, gps, tab)                                                  # :47-48
            ell += tab[lf.id]                                                          # :51
            grad = grad_tree(root, gps, n_hyp, leaf_weights=D[j, :])                   # :50,54
            hyp[j] = hyp[j] + step(hyp[j], grad)                                       # :55-56
        hist.append(ell)
        delta = abs(ell - np.mean(hist[-10:-1])) if it > 10 else np.inf                # :61
        c = c + 1 if delta < lam else 0                                                # :65-69
        if c >= 10:
            break
    for j, g in enumerate(gps):                                                        # :74-77 / :82-85
        g.kernel = ogp.make_kernel(leaves[j].kernel.kind, hyp[j][:-1])
        g.logNoise = float(hyp[j][-1])
        g.update_cholesky()
    return hyp, np.array(hist)
