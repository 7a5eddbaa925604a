"""Host-side mirror of the reference's model API over the HIP GP-expert path.

Names and argument meaning follow the Julia package (`!` dropped): buildDSMGP / buildPoE / buildBCM
(`src/treeStructure.jl:328-403`), fit / fit_naive (`src/fit.jl:67-122,294-304`), predict
(`src/common.jl:294-307`), update / infer / mll (`src/common.jl:323-355`, `src/optimize.jl:18-25`),
setparams / getparams (`src/optimize.jl:185-198`), train (`src/optimisers.jl:4-87`).

All per-leaf numerics (Gram, Cholesky, alpha, mll, predictive moments, gradients) are computed by
libdsmgp_hip.so through `hipabi.Context`; this module only holds the tree logic the reference keeps in
scalar Julia code (routing, log-domain mixture aggregation, weights), plus leaf sharding across ranks.
"""
import numpy as np

try:
    import xxhash as _xxhash
except ImportError:          # pragma: no cover
    _xxhash = None

from . import hipabi
from .kernels import IsoSE, ConstMean, KIND_ISO_SE, KIND_ARD_SE, KIND_ISO_LINEAR
from .tree import (DSMGPConfig, GPSumNode, build_tree, get_leaves, get_overlap, obs_table, share_schedule, share_decisions,
                   share_census, route, route_all, route_index, get_child, ordered_nodes, SHARE_COPY, SHARE_FULL, SHARE_PREFIX)
from . import dist as _dist

EPS = 1e-8  # `const ϵ` of src/DeepStructuredMixtures.jl:27


def _logsumexp(a, axis=None):
    """StatsFuns.logsumexp / `lse` (`src/common.jl:309-313`): max-shifted."""
    a = np.asarray(a, dtype=np.float64)
    if axis is None:
        m = np.max(a)
        m = m if np.isfinite(m) else 0.0
        return float(np.log(np.sum(np.exp(a - m))) + m)
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    return np.squeeze(np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True)) + m, axis=axis)


class TreeIndex:
    """Level-order index of the sum/split tree so that the bottom-up passes of `update!`/`infer!`/`mll`
    (`src/common.jl:323-355`, `src/optimize.jl:18-25`) and the top-down product of sum-node weights on a leaf's path
    run as one vector operation per tree level instead of a Python recursion per node (18k leaves at depth 4).
    Every sum node's `logweights` becomes a view into one flat edge array, so in-place updates are seen by the
    literal recursions too."""

    def __init__(self, root):
        nodes, depth, parent_edge = [root], [0], [-1]
        levels = []            # per depth d: internal nodes at d and their child edges (children sit at d + 1)
        frontier = [0]
        edge_child, edge_parent = [], []
        d = 0
        while frontier:
            par, seg, cnt, nxt = [], [], [], []
            e0 = len(edge_child)
            for ni in frontier:
                nd = nodes[ni]
                if nd.kind == "gp":
                    continue
                par.append(ni)
                seg.append(len(edge_child) - e0)
                cnt.append(len(nd.children))
                for c in nd.children:
                    nodes.append(c)
                    depth.append(d + 1)
                    parent_edge.append(len(edge_child))
                    edge_child.append(len(nodes) - 1)
                    edge_parent.append(ni)
                    nxt.append(len(nodes) - 1)
            if par:
                levels.append(dict(parent=np.array(par), seg=np.array(seg), cnt=np.array(cnt), e0=e0,
                                   e1=len(edge_child)))
            frontier = nxt
            d += 1
        self.nodes = nodes
        self.n = len(nodes)
        self.edge_child = np.array(edge_child, dtype=np.int64)
        self.edge_parent = np.array(edge_parent, dtype=np.int64)
        self.parent_edge = np.array(parent_edge, dtype=np.int64)
        is_sum = np.array([nd.kind == "sum" for nd in nodes])
        of_gps = np.array([nd.kind == "sum" and nd.of_gps for nd in nodes])
        nch = np.array([0 if nd.kind == "gp" else len(nd.children) for nd in nodes], dtype=np.float64)
        self.edge_is_sum = is_sum[self.edge_parent] if self.edge_parent.size else np.zeros(0, bool)
        self.edge_of_gps = of_gps[self.edge_parent] if self.edge_parent.size else np.zeros(0, bool)
        self.edge_prior = np.where(self.edge_is_sum, -np.log(np.maximum(nch[self.edge_parent], 1.0)), 0.0) \
            if self.edge_parent.size else np.zeros(0)
        self.leaf_node = np.array([i for i, nd in enumerate(nodes) if nd.kind == "gp"], dtype=np.int64)
        self.leaf_id = np.array([nodes[i].leaf for i in self.leaf_node], dtype=np.int64)
        for lv in levels:
            lv["parent_is_sum"] = is_sum[lv["parent"]]
        self.levels = levels
        self.lw = np.zeros(self.edge_child.size)       # log weight per edge (0 under split nodes)
        self.bind()

    def bind(self):
        """(Re)attach every sum node's logweights to the flat edge array, keeping the current values."""
        for lv in self.levels:
            for ni, a, k in zip(lv["parent"], lv["seg"] + lv["e0"], lv["cnt"]):
                nd = self.nodes[ni]
                if nd.kind == "sum":
                    self.lw[a:a + k] = nd._lw
                    nd._lw = self.lw[a:a + k]
        self.version = GPSumNode.assignments

    def bound(self):
        return self.version == GPSumNode.assignments

    def bottom_up(self, leaf_values, posterior="all"):
        """Node values of the mll recursion; posterior: "all" = update!, "gps" = infer!, None = leave weights."""
        val = np.empty(self.n)
        val[self.leaf_node] = leaf_values[self.leaf_id]
        for lv in reversed(self.levels):
            sl = slice(lv["e0"], lv["e1"])
            v = val[self.edge_child[sl]] + self.edge_prior[sl]
            m = np.maximum.reduceat(v, lv["seg"])
            m = np.where(np.isfinite(m), m, 0.0)
            z = m + np.log(np.add.reduceat(np.exp(v - np.repeat(m, lv["cnt"])), lv["seg"]))
            tot = np.add.reduceat(v, lv["seg"])
            val[lv["parent"]] = np.where(lv["parent_is_sum"], z, tot)
            if posterior is not None:
                post = v - np.repeat(z, lv["cnt"])
                if posterior == "gps":
                    post = np.where(self.edge_of_gps[sl], post, self.edge_prior[sl])
                self.lw[sl] = np.where(self.edge_is_sum[sl], post, 0.0)
        return val

    def weights_normalised(self, tol=1e-12):
        """Do the weights of every sum node add up to one?  Every path of the reference keeps them so (-log V at build,
        `src/treeStructure.jl:226`; log Dirichlet draws, `:260-286`; the posterior of `update!` / `infer!`,
        `src/common.jl:326-353`) -- but `logweights` is a plain field a caller may assign."""
        if not self.lw.size or not np.any(self.edge_is_sum):
            return True
        tot = np.bincount(self.edge_parent[self.edge_is_sum], weights=np.exp(self.lw[self.edge_is_sum]), minlength=self.n)
        par = np.unique(self.edge_parent[self.edge_is_sum])
        return bool(np.all(np.abs(tot[par] - 1.0) <= tol))

    def leaf_path_logweights(self):
        """log of the product of sum-node weights on every leaf's path, indexed by leaf id."""
        acc = np.zeros(self.n)
        for lv in self.levels:
            sl = slice(lv["e0"], lv["e1"])
            acc[self.edge_child[sl]] = acc[self.edge_parent[sl]] + self.lw[sl]
        out = np.zeros(self.leaf_id.size)
        out[self.leaf_id] = acc[self.leaf_node]
        return out


class Model:
    """Common state of DSMGP / PoE / gPoE / rBCM (`src/DeepStructuredMixtures.jl:108-130`)."""
    family = "dsmgp"

    def __init__(self, root, x, y, overlap, ctx=None, device=0, shard=None, n_sub=1, stream_budget=None):
        self.root = root
        self.x = np.asfortranarray(x, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        self.D = overlap  # leaf-overlap matrix (the reference calls this field D)
        self.leaves = get_leaves(root)
        self.L = len(self.leaves)
        self.shard = shard if shard is not None else _dist.Shard.single(self.L)
        self._ctx = ctx
        self._device = device
        self._n_sub = int(n_sub)      # > 1: that many concurrent contexts on this GPU (hipabi.MultiContext)
        self._stream_budget = stream_budget   # bytes (or "auto"): factor-and-discard over leaf groups (hipabi.StreamingContext)
        self.leaf_mll = np.full(self.L, np.nan)
        self.leaf_info = np.zeros(self.L, dtype=np.int32)
        self.last_fit_seconds = 0.0
        self.last_predict_seconds = 0.0
        self._uploaded = False
        self._schedule = None
        self._route_cache = None
        self._device_routing = False  # the context holds the tree (dsmgp_set_tree): predict routes its rows on the device
        self._tindex = None
        self.share_op = None          # per leaf: what the last fit did (SHARE_FULL / COPY / PREFIX)
        self.share_branch = None      # per leaf: the arm of the reference's fit! (tree.BRANCH_*)
        self.fit_census = None        # tree.share_census(share_branch): counts + the leaves of the arms computed in full here

    @property
    def tindex(self):
        if self._tindex is None:
            self._tindex = TreeIndex(self.root)
        elif not self._tindex.bound():      # somebody assigned fresh logweights arrays: take them over
            self._tindex.bind()
        return self._tindex

    @property
    def ctx(self):
        """GPU context, created on first use; raises if the HIP library or a GPU is missing."""
        if self._ctx is None:
            if self._stream_budget is not None:
                b = None if self._stream_budget == "auto" else int(self._stream_budget)
                self._ctx = hipabi.StreamingContext(self._device, b)
            elif self._n_sub > 1:
                self._ctx = hipabi.MultiContext(self._device, self._n_sub)
            else:
                self._ctx = hipabi.Context(self._device)
                if self.shard.world > 1:      # opt-in (DSMGP_EXCHANGE=rccl-device): the exchanges run through the library's
                    self.shard.device_comm(self._ctx)     # communicator; a collective every rank enters, or none
        return self._ctx

    # ---- kernel ids -> shared hyper-parameters --------------------------------------------------
    def kernel_table(self):
        """One (kernel, logNoise) per kernel id, taken from the first leaf carrying that id: the
        reference sets every leaf of an id to the same vector (`src/optimize.jl:188-198`)."""
        if getattr(self, "_ktab", None) is None:      # which leaf carries an id first is structural: cache it
            tab = {}
            for lf in self.leaves:
                tab.setdefault(lf.kernelid, lf)
            self._ktab = [tab[k] for k in sorted(tab)]
        return self._ktab

    def _push_hyper(self):
        for lf in self.kernel_table():
            hyp = np.concatenate([lf.kernel.loghyp(), [lf.logNoise]])
            self.ctx.set_hyper(lf.kernelid, lf.kernel.kind, hyp)

    def set_option(self, option, value):
        """`dsmgp_set_option` on the model's context.  An option that rebuilds the plan (OPT_FUSED_GRAM) drops the
        device-side test set with it: the route cache is told, so the next predict registers its rows again."""
        self.ctx.set_option(option, value)
        if self._route_cache is not None:
            self._route_cache["uploaded"] = False

    def _set_decisions(self, dec):
        """Record the sharing decisions of this fit (None: fit_naive!, every leaf in full) for the caller: which leaves
        the reference's default fit! would have sent through its (defective) row-deletion arm is part of the
        result, not only of the oracle."""
        if dec is None:
            dec = (np.zeros(self.L, dtype=np.int32), np.full(self.L, -1, dtype=np.int32), np.zeros(self.L, dtype=np.int64),
                   np.zeros(self.L, dtype=np.int8))
        self.share_op, self.share_branch = dec[0], dec[3]
        self.fit_census = share_census(dec[3])
        return dec

    # ---- leaf table ------------------------------------------------------------------------------
    def _upload(self, tau):
        loc = self.shard.local
        key = ("sched", None if tau is None else float(tau))
        if self._uploaded and self._schedule == key:
            return
        if len(loc) == 0:
            # a rank without leaves (fewer sharing groups than ranks, e.g. a small PoE on 8 GPUs) makes no device
            # call at all; it still joins every collective with empty contributions (_fit, _leaf_moments, ...)
            self._uploaded = True
            self._schedule = key
            self._route_cache = None
            if self.D is not None and tau is not None:
                self._set_decisions(share_decisions(self.leaves, self.D, tau))
            else:
                self._set_decisions(None)
            return
        if not self._uploaded:
            self.ctx.set_train(self.x, self.y)
        lv = [self.leaves[i] for i in loc]
        ptr, idx = obs_table(lv)
        self.ctx.set_leaves(ptr, idx, [lf.kernelid for lf in lv], [lf.mean.m for lf in lv])
        if self.D is not None and tau is not None:
            op, src, plen, _ = self._set_decisions(share_decisions(self.leaves, self.D, tau))
            g2l = {g: i for i, g in enumerate(loc)}
            lop = np.zeros(len(lv), dtype=np.int32)
            lsrc = np.full(len(lv), -1, dtype=np.int32)
            lpl = np.zeros(len(lv), dtype=np.int64)
            for i, g in enumerate(loc):
                if op[g] != SHARE_FULL and int(src[g]) in g2l:   # source must live on the same rank
                    lop[i], lsrc[i], lpl[i] = op[g], g2l[int(src[g])], plen[g]
            self.ctx.set_sharing(lop, lsrc, lpl)
        else:
            self._set_decisions(None)
            self.ctx.set_sharing(None, None, None)
        self._uploaded = True
        self._schedule = key
        self._route_cache = None      # set_leaves dropped the device-side test set
        self._register_tree(loc)

    def _register_tree(self, loc):
        """The routing of `predict` (`src/common.jl:101-122,181-196,275-292`) on the device: hand the context the tree as flat
        arrays, with every region's index in THIS rank's leaf table (-1: held elsewhere).  DSMGP models on a plain context;
        anything else (and a tree deeper than the device walk's stack) keeps routing on the host."""
        self._device_routing = False
        if self.family != "dsmgp" or self.root.kind == "gp" or not hasattr(self.ctx, "set_tree"):
            return
        ri = route_index(self.root)
        g2l = np.full(self.L, -1, dtype=np.int64)
        g2l[np.asarray(loc, dtype=np.int64)] = np.arange(len(loc))
        leaf_local = np.where(ri.leaf >= 0, g2l[np.maximum(ri.leaf, 0)], -1)
        try:
            self.ctx.set_tree(ri.kind, ri.first, ri.nchild, ri.sdim, ri.thr, leaf_local)
            self._device_routing = True
        except hipabi.DsmgpError:
            pass


class DSMGP(Model):
    family = "dsmgp"


class PoE(Model):
    family = "poe"


class gPoE(Model):
    family = "gpoe"


class rBCM(Model):
    family = "rbcm"


# ------------------------------------------------------------------------------------ builders

def build(x, y, K, V, eps, M, D, kernel, meanFun, logNoise, useSum, *, seed=7, device=0, ctx=None,
          cls=DSMGP, tau=0.05, fit_now=True, shard_world=None, n_sub=1, stream_budget=None):
    """`src/treeStructure.jl:405-437`. NOTE the reference swaps its positional K/V into the config:
    config.K (splits per split node) = V argument, config.V (children per sum node) = K argument."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    config = DSMGPConfig(meanFun, kernel, logNoise, M, V, K, D, eps, useSum)
    root = build_tree(x, np.asarray(y, dtype=np.float64), config, seed=seed)
    L = len(get_leaves(root))
    overlap = get_overlap(root, L)
    model = cls(root, x, y, overlap, ctx=ctx, device=device, n_sub=n_sub, stream_budget=stream_budget)
    if shard_world is not None:
        rank, world = shard_world
        op, src, _ = share_schedule(model.leaves, overlap, tau)
        model.shard = _dist.Shard.lpt([lf.nobs for lf in model.leaves], op, src, rank, world)
    if fit_now:
        fit(model, tau=tau)
    return model


def buildDSMGP(x, y, K, V, *, eps=0.5, M=30, D=2, kernel=None, meanFun=None, logNoise=1.0, sum=True, **kw):
    """buildDSMGP(x, y, K, V): K children under each sum node, V splits per split node
    (`src/treeStructure.jl:309-339`; README.md:51 calls it as buildDSMGP(x, y, 3, 4))."""
    kernel = IsoSE(1.0, 1.0) if kernel is None else kernel
    return build(x, y, K, V, eps, M, D, kernel, meanFun, logNoise, sum, cls=DSMGP, **kw)


def buildPoE(x, y, V, *, eps=0.0, M=30, D=2, kernel=None, meanFun, logNoise=1.0, generalized=False, **kw):
    """`src/treeStructure.jl:341-371` (meanFun is a required keyword there too)."""
    kernel = IsoSE(1.0, 1.0) if kernel is None else kernel
    return build(x, y, 1, V, eps, M, D, kernel, meanFun, logNoise, False, cls=gPoE if generalized else PoE, **kw)


def buildBCM(x, y, V, *, eps=0.0, M=30, D=2, kernel=None, meanFun=None, logNoise=1.0, robust=False, **kw):
    """`src/treeStructure.jl:373-403`: always the robust BCM."""
    kernel = IsoSE(1.0, 1.0) if kernel is None else kernel
    return build(x, y, 1, V, eps, M, D, kernel, meanFun, logNoise, False, cls=rBCM, **kw)


class GaussianProcess:
    """Single exact GP (`src/gaussianprocess.jl:14-80`) as a one-leaf table on the device."""

    def __init__(self, x, y, *, mean=None, kernel=None, logNoise=np.log(7.0), run_cholesky=False, device=0, ctx=None):
        from .tree import GPNode
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        y = np.asarray(y, dtype=np.float64)
        mean = ConstMean(float(np.mean(y))) if mean is None else mean
        kernel = IsoSE(0.0, 0.0) if kernel is None else kernel
        N, D = x.shape
        leaf = GPNode(np.arange(N), np.full(D, -np.inf), np.full(D, np.inf), kernel, 0, mean, logNoise)
        leaf.leaf = 0
        self.node = leaf
        self.model = Model(leaf, x, y, None, ctx=ctx, device=device)
        self.N, self.D = N, D
        if run_cholesky:
            update_cholesky(self)

    @property
    def kernel(self):
        return self.node.kernel

    @property
    def logNoise(self):
        return self.node.logNoise


def update_cholesky(gp):
    """`update_cholesky!(gp)` (`src/gaussianprocess.jl:82-108`)."""
    fit_naive(gp.model)
    return gp


def prediction(gp, xtest):
    """`prediction(gp, xtest)` -> (mu, diag of Sigma) (`src/gaussianprocess.jl:110-137`; only the
    diagonal of Sigma is ever consumed, `src/common.jl:136,147`)."""
    xt = _test_matrix(gp.model, xtest)
    if xt.shape[0] == 0:
        return np.zeros(0), np.zeros(0)
    # registered like the test set of a tree model (cached by content): a loop of update_cholesky! + prediction on the same
    # rows carries them through the factorisation launches from its second pass on, and prediction only finishes the moments
    rc = _routing(gp.model, xt)
    mu, var = _leaf_moments(gp.model, xt, rc)
    return mu, var


# ------------------------------------------------------------------------------------ fit

_RANK_FAILED = -(1 << 30)      # info value of every leaf of a rank whose fit raised (LAPACK's info is >= 0 here)


def _fit(model, tau):
    model._upload(tau)
    if model.shard.world > 1 and _ctx_type(model) is hipabi.Context:
        _ = model.ctx          # every rank, also one without leaves, joins the set-up of the device exchange (collective)
    failure = None
    if len(model.shard.local) == 0:
        mll_loc, info_loc, sec = np.zeros(0), np.zeros(0, dtype=np.int32), 0.0
    elif model.shard.world == 1 or model.shard.comm_ctx is not None:     # (the opt-in device exchange reads the fit's results
        model._push_hyper()                                              # in HBM: nothing to send for a fit that raised)
        mll_loc, info_loc, sec = model.ctx.fit()
    else:
        # One of several ranks: whatever goes wrong HERE (out of memory, a HIP error, bad hyper-parameters on this shard) must not
        # keep this rank out of the collective below while the others wait in it.  The failure travels as a sentinel in the info
        # column -- every rank learns of it from the same gather and raises; nobody is left in a later collective alone.
        try:
            model._push_hyper()
            mll_loc, info_loc, sec = model.ctx.fit()
        except Exception as e:      # noqa: BLE001
            failure = e
            mll_loc = np.full(len(model.shard.local), np.nan)
            info_loc, sec = np.full(len(model.shard.local), _RANK_FAILED, dtype=np.int32), 0.0
    cols = np.stack([mll_loc, info_loc.astype(np.float64)], axis=1)
    if model.shard.comm_ctx is not None:
        both = model.shard.fit_exchange(model.ctx, cols)     # device to device over RCCL, then one copy to the host
    else:
        both = model.shard.gather_leaf_columns(cols)         # one collective
    model.leaf_mll = np.ascontiguousarray(both[:, 0])
    model.leaf_info = both[:, 1].astype(np.int32)
    model.last_fit_seconds = sec
    if failure is not None:
        raise failure
    lost = np.flatnonzero(model.leaf_info == _RANK_FAILED)
    if lost.size:
        raise RuntimeError(f"fit failed on rank {int(model.shard.owner[lost[0]])} (its {lost.size} leaves have no result); "
                           "the error is in that rank's output")
    bad = np.flatnonzero(model.leaf_info != 0)
    if bad.size:
        raise np.linalg.LinAlgError(f"leaf {int(bad[0])}: leading minor of order {int(model.leaf_info[bad[0]])} "
                                    "is not positive definite")
    return sec


class FitSeconds(float):
    """What `fit!` returns -- the seconds of the leaf loop (`src/fit.jl:88,121`) -- carrying the census of the sharing
    decisions: `.census` = dict(full, copy, prefix, lowrank_as_full, leading_as_full, lowrank_leaves, leading_leaves)
    over the arms of the REFERENCE's fit!.  `lowrank_leaves` are the leaves whose reference result comes from its
    row-deletion arm (`src/fit.jl:174-201`; numerically defective) and is an exact factorisation here."""
    census = None


def fit(model, tau=0.05):
    """`fit!(model; τ)`: shared-Cholesky fit (copy + prefix sharing). Returns seconds (`FitSeconds`: the
    census of the sharing decisions rides on it, and stays on the model as `model.fit_census`)."""
    target = model.model if isinstance(model, GaussianProcess) else model
    out = FitSeconds(_fit(target, tau))
    out.census = target.fit_census
    return out


def fit_naive(model):
    """`fit_naive!`: one full factorisation per leaf (`src/fit.jl:294-304`)."""
    target = model.model if isinstance(model, GaussianProcess) else model
    return _fit(target, None)


# ------------------------------------------------------------------------------------ mll / weights

def mll(model):
    """Tree log marginal likelihood (`src/optimize.jl:18-25`)."""
    if isinstance(model, GaussianProcess):
        return float(model.model.leaf_mll[0])
    return float(model.tindex.bottom_up(model.leaf_mll, posterior=None)[0])


def _mll_recursive(model):
    """The literal recursion of `src/optimize.jl:18-25` (cross-check of TreeIndex.bottom_up in the CPU tests)."""

    def rec(node):
        if node.kind == "gp":
            return float(model.leaf_mll[node.leaf])
        if node.kind == "split":
            return sum(rec(c) for c in node.children)
        K = len(node.children)
        return float(_logsumexp(np.array([-np.log(K) + rec(c) for c in node.children])))

    return rec(model.root)


def mll_table(model):
    """`mll!(spn, L)`: value per node id (`src/optimize.jl:27-39`)."""
    tab = {}

    def rec(node):
        if node.kind == "gp":
            v = float(model.leaf_mll[node.leaf])
        elif node.kind == "split":
            v = sum(rec(c) for c in node.children)
        else:
            K = len(node.children)
            v = float(_logsumexp(np.array([-np.log(K) + rec(c) for c in node.children])))
        tab[node.id] = v
        return v

    rec(model.root)
    return tab


def update(model):
    """`update!(model)`: posterior sum-node weights (`src/common.jl:323-334`). Returns the root value."""
    return float(model.tindex.bottom_up(model.leaf_mll, posterior="all")[0])


def _update_recursive(model):
    """The literal recursion of `src/common.jl:323-334` (cross-check of the level-order pass in the CPU tests)."""

    def rec(node):
        if node.kind == "gp":
            return float(model.leaf_mll[node.leaf])
        if node.kind == "split":
            return sum(rec(c) for c in node.children)
        K = len(node.children)
        lw = np.array([-np.log(K) + rec(c) for c in node.children])
        z = float(_logsumexp(lw))
        node.logweights = lw - z
        return z

    return rec(model.root)


def infer(model):
    """`infer!(model)` (`src/common.jl:336-355`): only sums over GPs keep posterior weights."""
    return float(model.tindex.bottom_up(model.leaf_mll, posterior="gps")[0])


def reset_weights(model):
    """`reset_weights!` (`src/common.jl:357-363`)."""
    for n in ordered_nodes(model.root):
        if n.kind == "sum":
            n.logweights[:] = -np.log(len(n.children))


# ------------------------------------------------------------------------------------ parameters

def getparams(model):
    """Concatenated log-scale hyper-vector per kernel id: [logl..., logs, logNoise]."""
    return np.concatenate([np.concatenate([lf.kernel.loghyp(), [lf.logNoise]]) for lf in model.kernel_table()])


def setparams(model, hyp):
    """`setparams!(spn, hyp)` (`src/optimize.jl:188-198`, `src/gaussianprocess.jl:153-161`)."""
    hyp = np.asarray(hyp, dtype=np.float64)
    tab = model.kernel_table()
    c = 0
    chunks = {}
    for lf in tab:
        n = lf.kernel.nparams() + 1
        chunks[lf.kernelid] = hyp[c:c + n]
        c += n
    if c != hyp.size:
        raise ValueError(f"hyper-vector has {hyp.size} entries, model needs {c}")
    for lf in model.leaves:
        h = chunks[lf.kernelid]
        lf.logNoise = float(h[-1])
        lf.kernel.set_loghyp(h[:-1])


# ------------------------------------------------------------------------------------ gradients / training

def updategradients(model, active=None):
    """`updategradients!(spn)` (`src/fit.jl:306-311`): per-leaf gradient vectors in the reference's order
    [dl..., ds, dnoise] (`src/gaussianprocess.jl:212-214`), computed on the device for the local leaves and
    gathered.  Also stored on the leaves (kernel.dl / kernel.ds / dnoise) like the reference does.
    `active` (one flag per leaf, default all): only those leaves' gradients are computed, the other rows are zero --
    what `finetune!` needs, whose pass for leaf j weights leaf l's gradient by the overlap D[j, l] (`src/optimize.jl:101`).
    The gradients stored on the leaves (kernel.dl / kernel.ds / dnoise) are ZERO outside the active set, where the reference
    stores every leaf's gradient on every pass (they are multiplied by D[j, l] = 0 there and never read otherwise); the
    streaming context ignores the mask and computes all of them."""
    target = model.model if isinstance(model, GaussianProcess) else model
    stride = max(lf.kernel.nparams() + 1 for lf in target.leaves)
    if len(target.shard.local) and (active is not None or getattr(target, "_grad_masked", False)):
        target.ctx.set_gradient_leaves(None if active is None else np.asarray(active)[target.shard.local])
        target._grad_masked = active is not None
    g_loc = target.ctx.gradients(stride) if len(target.shard.local) else np.zeros((0, stride))
    g = target.shard.gather_leaf_columns(g_loc[:, :stride])
    for lf, row in zip(target.leaves, g):
        n = lf.kernel.nparams()
        if lf.kernel.kind == KIND_ISO_LINEAR:
            lf.kernel.dl = float(row[0])
        else:
            lf.kernel.dl = row[: n - 1].copy() if lf.kernel.kind == KIND_ARD_SE else float(row[0])
            lf.kernel.ds = float(row[n - 1])
        lf.dnoise = float(row[n])
    target.leaf_grad = g
    return g


def grad_mll(model, leaf_weights=None):
    """`∇mll!(spn, 0.0, 0.0, L, L[root], grad)` (`src/optimize.jl:42-89`): gradient of the tree log marginal
    w.r.t. the shared hyper-vector (concatenated per kernel id under sums over GPs).  With `leaf_weights` (one factor
    per leaf: a row of the overlap matrix) it is the `finetune!` variant, `src/optimize.jl:91-150`."""
    if isinstance(model, GaussianProcess):
        return model.model.leaf_grad[0][: model.node.kernel.nparams() + 1].copy()
    tab = mll_table(model)
    logS = tab[model.root.id]
    n_hyp = getparams(model).size
    grad = np.zeros(n_hyp)

    def rec(node, dparent, lrho, g):
        if node.kind == "gp":
            w = np.exp(-logS + lrho + tab[node.id] + dparent)                 # :48
            if leaf_weights is not None:
                w = w * leaf_weights[node.leaf]                               # :101
            g += model.leaf_grad[node.leaf][: g.size] * w                     # :49
        elif node.kind == "split":
            for c in node.children:
                rec(c, dparent + (tab[node.id] - tab[c.id]), lrho, g)         # :58-61
        elif node.of_gps:
            c0 = 0
            for c in node.children:                                           # :76-89
                nn = c.kernel.nparams() + 1
                rec(c, dparent, lrho, g[c0:c0 + nn])
                c0 += nn
        else:
            K = len(node.children)
            for c in node.children:
                rec(c, -np.log(K) + dparent, np.log(K) + lrho, g)             # :70-73

    rec(model.root, 0.0, 0.0, grad)
    return grad


class ADAM:
    """Flux.Optimise.ADAM stand-in for `train!`.  With stateful=False (default) the moment estimates are
    reset every step, which is what the reference effectively runs: `hyp += grad` rebinds `hyp`, so Flux's
    IdDict-keyed state never carries over and the step is eta*g/(|g|+eps) (`src/optimisers.jl:78-79`,
    SURVEY F9).  stateful=True is the textbook optimiser."""

    def __init__(self, eta=1e-3, beta=(0.9, 0.999), eps=1e-8, stateful=False):
        self.eta, self.beta, self.eps, self.stateful = eta, beta, eps, stateful
        self.m = self.v = None
        self.t = 0

    def apply(self, x, g):
        if not self.stateful or self.m is None:
            self.m = np.zeros_like(g)
            self.v = np.zeros_like(g)
            self.t = 0
        b1, b2 = self.beta
        self.t += 1
        self.m = b1 * self.m + (1 - b1) * g
        self.v = b2 * self.v + (1 - b2) * g * g
        return self.m / (1 - b1 ** self.t) / (np.sqrt(self.v / (1 - b2 ** self.t)) + self.eps) * self.eta


class RMSProp:
    """Flux.Optimise.RMSProp stand-in (default optimiser of `train!(gp)`, `src/optimisers.jl:91`); stateful=False
    restarts the running average every step, which is what the reference's rebinding `hyp += grad` amounts to
    (SURVEY F9): the step is eta * g / (sqrt(1 - rho) |g| + eps)."""

    def __init__(self, eta=1e-3, rho=0.9, eps=1e-8, stateful=False):
        self.eta, self.rho, self.eps, self.stateful = eta, rho, eps, stateful
        self.acc = None

    def apply(self, x, g):
        if not self.stateful or self.acc is None:
            self.acc = np.zeros_like(g)
        self.acc = self.rho * self.acc + (1 - self.rho) * g * g
        return g * (self.eta / (np.sqrt(self.acc) + self.eps))


def _train_gp(gp, optim, iterations, lam, randinit, seed, verbose):
    """`train!(gp::GaussianProcess; iterations, optim, λ)` (`src/optimisers.jl:89-145`): ascent on one GP's log
    marginal; a NaN log marginal (or a factorisation that fails: LAPACK info > 0, which the reference's potrf! call
    ignores and which then shows up as NaN) rolls back to the previous hyper-vector and returns; early stop when the
    last value is within λ of the mean of the nine before it (`:118`, a single hit suffices here)."""
    from .datagen import normal
    target = gp.model
    n = getparams(target).size
    hyp = normal(seed, 0, n) if randinit else getparams(target).copy()
    old = hyp.copy()
    hist = []
    for it in range(1, iterations + 1):
        setparams(target, hyp)
        try:
            update_cholesky(gp)
            ell = mll(gp)
        except (np.linalg.LinAlgError, ValueError, hipabi.DsmgpError):   # failed potrf / non-finite hyper-parameters
            ell = float("nan")
        hist.append(ell)
        if np.isnan(ell):                                                     # :115-119
            setparams(target, old)
            update_cholesky(gp)
            return gp, np.array(hist)
        delta = abs(ell - np.mean(hist[-10:-1])) if it > 10 else np.inf        # :121
        if verbose:
            print(f"iter {it}: mll {ell:.6f} delta {delta:.3g}")
        if delta < lam:                                                       # :125-128
            return gp, np.array(hist)
        updategradients(gp)
        g = grad_mll(gp)
        old = hyp.copy()
        hyp = hyp + optim.apply(hyp, g)                                       # :135-137 (ascent)
    setparams(target, hyp)
    update_cholesky(gp)
    return gp, np.array(hist)


def train(model, optim=None, *, iterations=10_000, lam=None, randinit=True, earlystop=10, seed=0, tau=0.05, verbose=False):
    """`train!(model, optim; iterations, λ, randinit, earlystop)` (`src/optimisers.jl:4-87`): gradient ASCENT
    on the tree log marginal over one shared hyper-vector.  Returns (model, history of root mll).
    For a single `GaussianProcess` it is `train!(gp; iterations, optim, λ)` (`src/optimisers.jl:89-145`:
    RMSProp, λ = 0.1, rollback on a NaN log marginal)."""
    from .datagen import normal
    if isinstance(model, GaussianProcess):
        return _train_gp(model, RMSProp() if optim is None else optim, iterations, 0.1 if lam is None else lam, randinit, seed,
                         verbose)
    lam = 0.05 if lam is None else lam
    optim = ADAM() if optim is None else optim
    n = getparams(model).size
    hyp = normal(seed, 0, n) if randinit else getparams(model)
    hist = []
    c = 0
    has_leaves = len(model.shard.local) > 0
    if has_leaves:
        model.ctx.set_joint(False)     # fit is not followed by predict inside the loop: keep resident test rows out of it
    # factor-and-discard context: a pass over the leaf groups cannot be revisited, so the loop's fit! asks for the
    # gradients of the same pass up front (one pass per iteration instead of a fit pass plus a fit + gradient pass)
    streaming = has_leaves and hasattr(model.ctx, "want_gradients")
    if streaming:
        model.ctx.want_gradients = max(lf.kernel.nparams() + 1 for lf in model.leaves)
        model.ctx.groups = None
    try:
        return _train_loop(model, optim, hyp, hist, c, iterations, lam, earlystop, tau, verbose, streaming)
    finally:
        if streaming:                  # also after an error inside the loop: later passes must not collect gradients
            model.ctx.want_gradients = 0
            model.ctx.groups = None
        if has_leaves:
            model.ctx.set_joint(True)


def _train_loop(model, optim, hyp, hist, c, iterations, lam, earlystop, tau, verbose, streaming=False):
    def plain_fits():
        if streaming:                  # the fits after the loop need no gradients: smaller groups, fewer passes
            model.ctx.want_gradients = 0
            model.ctx.groups = None

    for it in range(1, iterations + 1):
        setparams(model, hyp)
        fit(model, tau=tau)
        ell = mll(model)
        hist.append(ell)
        delta = abs(ell - np.mean(hist[-10:-1])) if it > 10 else np.inf       # :53
        c = c + 1 if delta < lam else 0
        if verbose:
            print(f"iter {it}: mll {ell:.6f} delta {delta:.3g}")
        if c >= earlystop:
            plain_fits()
            return model, np.array(hist)
        updategradients(model)
        g = grad_mll(model)
        hyp = hyp + optim.apply(hyp, g)                                       # :78-79 (ascent)
    plain_fits()
    setparams(model, hyp)
    fit(model, tau=tau)
    return model, np.array(hist)


def _overlap_row(D, j):
    return D.row(j) if hasattr(D, "row") else np.asarray(D[j, :], dtype=np.float64)


def finetune(model, optim=None, *, iterations=1000, lam=0.5, tau=0.05, verbose=False):
    """`finetune!(model, optim; iterations, λ)` (`src/finetuning.jl:3-87`): one hyper-vector PER LEAF.  Every
    iteration visits every leaf j: all leaves are set to leaf j's vector, the whole tree is refitted, and leaf j's
    vector takes an ascent step along the tree gradient in which leaf l's contribution is weighted by the overlap
    D[j, l] (`src/optimize.jl:91-150`; the diagonal of D is zero, so the leaf's own term does not enter -- the
    reference passes `D` at `:54`, not the `Dd` it prepares at `:30-31`).  That is L whole-tree fit! +
    updategradients! passes per iteration, each one batched call here.  The history is the sum over leaves of each
    leaf's own log marginal at its own vector (`:51,59`); early stopping as `:61-79`.  At the end every leaf keeps
    its own vector: the leaves get kernel ids of their own and are refactorised (`:74-77,82-85`).  One kernel id only: the reference's
    `setparams!(spn, hyp_)` with a single leaf's vector is ill-formed for kernel vectors."""
    if len(model.kernel_table()) != 1:
        raise NotImplementedError("finetune: models with a kernel vector are not supported (ill-formed in the reference)")
    optim = ADAM() if optim is None else optim
    L = model.L
    hyp = [np.concatenate([lf.kernel.loghyp(), [lf.logNoise]]) for lf in model.leaves]
    rows = [_overlap_row(model.D, j) for j in range(L)]
    hist, c = [], 0
    model.ctx.set_joint(False)
    try:
        for it in range(1, iterations + 1):
            ell = 0.0
            for j, lf in enumerate(model.leaves):
                setparams(model, hyp[j])
                fit(model, tau=tau)
                updategradients(model, active=rows[j] != 0)       # every other leaf's term is multiplied by D[j, l] = 0
                ell += float(model.leaf_mll[lf.leaf])                             # :51
                hyp[j] = hyp[j] + optim.apply(hyp[j], grad_mll(model, leaf_weights=rows[j]))   # :54-56
            hist.append(ell)
            delta = abs(ell - np.mean(hist[-10:-1])) if it > 10 else np.inf          # :61
            c = c + 1 if delta < lam else 0
            if verbose:
                print(f"iter {it}: sum of leaf mll {ell:.6f} delta {delta:.3g}")
            if c >= 10:
                break
    finally:
        model.ctx.set_joint(True)
        if getattr(model, "_grad_masked", False) and len(model.shard.local):
            model.ctx.set_gradient_leaves(None)
            model._grad_masked = False
    # every leaf keeps its own hyper-parameters: a kernel id per leaf, then one factorisation each (:74-77, :82-85)
    before = [(lf.kernelid, lf.logNoise, lf.kernel.loghyp().copy()) for lf in model.leaves]
    for j, lf in enumerate(model.leaves):
        lf.kernelid = j
        lf.logNoise = float(hyp[j][-1])
        lf.kernel.set_loghyp(hyp[j][:-1])
    model._ktab = None
    model._uploaded = False
    try:
        fit_naive(model)
    except Exception:
        # leave the model as it was before the per-leaf ids were assigned (one shared vector, refittable)
        for lf, (kid, ln, lh) in zip(model.leaves, before):
            lf.kernelid, lf.logNoise = kid, ln
            lf.kernel.set_loghyp(lh)
        model._ktab = None
        model._uploaded = False
        raise
    return model, np.array(hist)


# ------------------------------------------------------------------------------------ predict

def _content_hash(a):
    """64-bit hash of an array's contents: what tells `predict` that it is handed the test set it has registered (the cache key
    is (shape, this hash): a 64-bit content hash decides whether the rows are registered again -- a collision between two test
    matrices of one shape would reuse the first one's routes, at odds of 2^-64 per pair).  xxh3 over the array's own buffer in
    whichever contiguous layout it has -- `predict` hands over Fortran-ordered matrices (the layout of the C ABI), whose buffer
    is the C-ordered buffer of the transpose; anything else (a strided slice) is hashed through one contiguous copy.  xxhash is
    an optional dependency: without it the bytes go through Python's own hash (a copy and a SipHash: 0.65 ms for 10k x 8 rows
    against 0.03 ms)."""
    if a.flags.c_contiguous:
        buf = a
    elif a.flags.f_contiguous:
        buf = a.T
    else:
        buf = np.ascontiguousarray(a)
    if a.size == 0:
        return 0
    if _xxhash is not None:
        return _xxhash.xxh3_64_intdigest(memoryview(buf).cast("B"))
    return hash(buf.tobytes())


def _routing(model, xt, host_routes=True):
    """Routes of one test set, cached on the model: which rows each leaf predicts (CSR over all leaves and over
    this rank's leaves) and, filled lazily by the aggregation, the child masks of every split node.  With
    host_routes=False only the cache entry is made: the context routes the rows itself (`set_test_routed`)."""
    key = (xt.shape, _content_hash(xt))
    rc = model._route_cache
    if rc is None or rc["key"] != key:
        rc = model._route_cache = dict(key=key, ptr=None, idx=None, lptr=None, lidx=None, masks={}, uploaded=False)
    if host_routes and rc["ptr"] is None:
        ptr, idx = route(model.root, xt) if model.family == "dsmgp" else route_all(model.root, xt.shape[0])
        loc = np.asarray(model.shard.local, dtype=np.int64)
        if loc.size == ptr.size - 1 and np.array_equal(loc, np.arange(loc.size)):
            lptr, lidx = ptr, idx                          # this rank holds every leaf, in order
        else:                                              # the segments of this rank's leaves, one after the other (no loop
            cnt = (ptr[1:] - ptr[:-1])[loc]                # over leaves: 18k of them at depth 4)
            lptr = np.zeros(loc.size + 1, dtype=np.int64)
            np.cumsum(cnt, out=lptr[1:])
            lidx = idx[np.repeat(ptr[loc] - lptr[:-1], cnt) + np.arange(lptr[-1])] if loc.size else np.zeros(0, np.int64)
        rc.update(ptr=ptr, idx=idx, lptr=lptr, lidx=lidx)
    return rc


def _register_rows(model, xt, rc):
    """The test rows -> the context (once per test matrix): routed on the device where the context holds the tree."""
    if rc["uploaded"]:
        return
    if model._device_routing:
        try:
            model.ctx.set_test_routed(xt)
            rc["uploaded"] = True
            return
        except hipabi.DsmgpError as e:
            # the routing workspace (bitmap + prefixes: 8 L ceil(n_t / 32) bytes) did not fit: the host lists need none of it.
            # Anything else (a row outside a split region: DsmgpDomainError, a ValueError like the host routing's) is the caller's
            if e.code != hipabi.E_NOMEM:
                raise
    if rc["ptr"] is None:
        _routing(model, xt)
    model.ctx.set_test(xt, rc["lptr"], rc["lidx"])
    rc["uploaded"] = True


def _leaf_moments(model, xt, rc):
    """(mu, var) per (leaf, routed row) for ALL leaves, computed on the owning ranks."""
    counts = np.diff(rc["ptr"])
    if len(model.shard.local) == 0:
        return model.shard.gather_ragged_pair(np.zeros(0), np.zeros(0), counts)
    if not rc["uploaded"]:
        model.ctx.set_test(xt, rc["lptr"], rc["lidx"])      # (the per-(leaf, row) moments come back aligned with THESE lists)
        rc["uploaded"] = True
    model.last_predict_seconds = model.ctx.predict_run()
    mu_l, var_l = model.ctx.predict_fetch()
    return model.shard.gather_ragged_pair(mu_l, var_l, counts)


def _test_matrix(model, xtest):
    """`xtest` as the n_t x D Fortran-ordered matrix the C ABI reads; a matrix of another width than the training data is
    refused here, on every path (device routing reads n_t * D doubles from the caller's buffer)."""
    xt = np.asfortranarray(xtest, dtype=np.float64)
    if xt.ndim == 1:
        xt = xt.reshape(-1, 1)
    if xt.ndim != 2 or xt.shape[1] != model.x.shape[1]:
        raise ValueError(f"test matrix of shape {xt.shape}: the model was trained on D = {model.x.shape[1]} columns")
    return xt


def resident_test(model, xtest, tau=0.05):
    """Register `xtest` as the resident test set BEFORE the next fit (no reference counterpart): that fit then
    carries the test rows through its factorisation launches and the following `predict(model, xtest)` only
    finishes the moments.  `predict` registers its argument by itself, which pays off from the second fit on;
    this call is for evaluation loops that know their test set up front and for the streaming context, where a
    prediction after an unprepared fit costs a second pass over all leaf groups."""
    xt = _test_matrix(model, xtest)
    if xt.shape[0] == 0:
        return                      # nothing to carry along
    model._upload(tau)
    rc = _routing(model, xt, host_routes=not model._device_routing)
    if len(model.shard.local):
        _register_rows(model, xt, rc)


def predict(model, xtest):
    """`predict(model, x)` -> (mu, var) of length n_t (`src/common.jl:294-307`).  The rows are registered with the context once per
    test matrix: whether `x` is the registered one is decided by (shape, 64-bit content hash) -- `_content_hash`.  The per-(leaf, row) moments stay on
    the device and are aggregated there (`dsmgp_aggregate*`); contexts without that entry (the streaming context, whose
    moments are on the host anyway) use the host rules below."""
    if isinstance(model, GaussianProcess):
        mu, var = prediction(model, xtest)
        var = np.where(var <= 0, EPS, var)
        return mu, var
    xt = _test_matrix(model, xtest)
    model._scores_on_device = False
    if xt.shape[0] == 0:            # no rows: (Float64[], Float64[]) as in the reference; no device call, no collective (every rank sees the same x)
        return np.zeros(0), np.zeros(0)
    if model.family == "dsmgp" and model.root.kind != "gp" and not model.tindex.weights_normalised():
        # `_predict` shifts the means by c = mu_min - 1 before it weighs them (`src/common.jl:134-143,275-302`): with weights
        # that add up to one the shift cancels and the recursion IS the flat mixture the device aggregates; with weights a
        # caller assigned by hand it leaves c (1 - sum of weights) per sum node behind.  Then: the literal recursion on the
        # host, from the per-(leaf, row) moments of the device (the same on every rank: it depends on the weights alone)
        rc = _routing(model, xt)
        mu, var = _leaf_moments(model, xt, rc)
        return _aggregate_dsmgp(model, xt, rc["ptr"], mu, var)
    if hasattr(_ctx_type(model), "aggregate_partial"):       # decided by the model's construction: the same on every rank
        # (a rank that holds leaves and whose context holds the tree routes its rows on the device: no host lists at all)
        return _predict_device(model, xt, _routing(model, xt, host_routes=not model._device_routing))
    rc = _routing(model, xt)
    mu, var = _leaf_moments(model, xt, rc)
    if model.family == "dsmgp":
        return _aggregate_dsmgp_flat(model, xt.shape[0], rc, mu, var)
    return _aggregate_poe(model, xt, rc["ptr"], mu, var)


def _ctx_type(model):
    """Type of the model's device context without creating it (a rank that owns no leaves never does)."""
    if model._ctx is not None:
        return type(model._ctx)
    if model._stream_budget is not None:
        return hipabi.StreamingContext
    return hipabi.MultiContext if model._n_sub > 1 else hipabi.Context


def _aggregation_spec(model):
    """(family, leaf_coef, leaf_group, n_groups, plain, prior_leaf) of `dsmgp_aggregate` for this model."""
    root = model.root
    if model.family == "dsmgp":
        return hipabi.AGG_MIXTURE, np.exp(model.tindex.leaf_path_logweights()), None, 0, root.kind == "gp", None
    if root.kind == "gp" or model.family == "poe":
        return hipabi.AGG_POE, np.ones(model.L), None, 0, False, None
    if model.family == "gpoe":
        return hipabi.AGG_GPOE, np.full(model.L, 1.0 / len(root.children)), None, 0, False, None      # src/common.jl:215
    group = np.zeros(model.L, dtype=np.int32)
    for g, c in enumerate(root.children):                                                                 # :231
        for lf in get_leaves(c):
            group[lf.leaf] = g
    return hipabi.AGG_RBCM, None, group, len(root.children), False, model.leaves[0]


def _finish_partial(model, xt, family, part, n_groups, plain, prior_leaf):
    """Host form of `agg_finish_kernel` on summed partial sums (several ranks or contexts hold the leaves)."""
    if family == hipabi.AGG_MIXTURE:
        m = part[0]
        return m, (part[2] if plain else part[2] + (part[1] - m * m))
    if family != hipabi.AGG_RBCM:
        return part[0] / part[1], 1.0 / part[1]
    s = _prior_diag(prior_leaf, xt) + np.exp(2 * prior_leaf.logNoise)
    Cc = 1.0 / s
    m = np.zeros(xt.shape[0])
    for g in range(n_groups):
        T = part[2 * g + 1]
        seen = T != 0.0                      # no leaf of this child saw the row: the group is skipped (agg_finish_kernel)
        Ts = np.where(seen, T, 1.0)
        M = part[2 * g] / Ts
        beta = 0.5 * (np.log(s) - np.log(1.0 / Ts))
        Cc = Cc + np.where(seen, (beta * Ts) - (beta / s), 0.0)
        m = m + np.where(seen, M * (beta * Ts), 0.0)
    return m / Cc, 1.0 / Cc


def _predict_device(model, xt, rc):
    family, coef, group, G, plain, prior = _aggregation_spec(model)
    loc = model.shard.local
    have = len(loc) > 0
    if have:
        _register_rows(model, xt, rc)
        model.last_predict_seconds = model.ctx.predict_run()
    single = model.shard.world == 1 and isinstance(model.ctx, hipabi.Context) and model.shard.comm_ctx is None
    if single:      # one context holds every leaf: partial sums, finish and (later) scores never leave the device
        mu, var = model.ctx.aggregate(family, coef, group, G, plain=plain, prior_kernel_id=prior.kernelid if prior else 0)
        model._scores_on_device = True
        return mu, var
    if model.shard.comm_ctx is not None:
        # the exchange step of predict inside the library: partial sums stay in HBM, are all-gathered over RCCL on the
        # context's stream and added in rank order on the device; the finish runs on the total
        W = hipabi.agg_width(family, G)
        if have:
            model.ctx.aggregate_partial(family, None if coef is None else coef[loc], None if group is None else group[loc], G,
                                        fetch=False)
            model.ctx.aggregate_exchange(W)
            mu, var = model.ctx.aggregate_finish(None, plain=plain, prior_kernel_id=prior.kernelid if prior else 0)
            model._scores_on_device = True
            return mu, var
        part = model.ctx.aggregate_exchange_empty(W, xt.shape[0])
        return _finish_partial(model, xt, family, part, G, plain, prior)
    if have:
        part = model.ctx.aggregate_partial(family, None if coef is None else coef[loc], None if group is None else group[loc], G)
    else:
        part = np.zeros((hipabi.agg_width(family, G), xt.shape[0]))
    part = model.shard.allgather_sum(part)          # the exchange step of predict: W x n_t doubles per rank
    return _finish_partial(model, xt, family, part, G, plain, prior)


def scores(model, y_test, mu=None, var=None):
    """dict(mse, sse, mae, sae, nlpd) (`src/scorefunctions.jl:6-16`) of the last `predict(model, x)`: computed on the
    device from the aggregated prediction still resident there when one context holds the model, else from (mu, var)."""
    y_test = np.ascontiguousarray(y_test, dtype=np.float64)
    if getattr(model, "_scores_on_device", False) and mu is None:
        return model.ctx.scores(y_test)
    if mu is None or var is None:
        raise ValueError("scores: pass the (mu, var) that predict returned (they are not resident on one device)")
    return dict(mse=mse(y_test, mu), sse=sse(y_test, mu), mae=mae(y_test, mu), sae=sae(y_test, mu), nlpd=nlpd(y_test, mu, var))


def _aggregate_dsmgp_flat(model, n_t, rc, mu, var):
    """The nested log-domain recursion of `_predict` (`src/common.jl:275-302`) is linear in the leaf quantities
    (mu - c, mu^2, sigma^2): unrolled, a test row's prediction is the flat mixture over the leaves it visits with
    weight = product of the sum-node weights on the leaf's path (split nodes only route).  So
        mu = sum_l W_l mu_l,   v = sum_l W_l sigma2_l + sum_l W_l mu_l^2 - mu^2
    in three weighted bincounts over the (leaf, row) entries.  `_aggregate_dsmgp` below is the literal recursion,
    kept as the cross-check (tests/test_host_cpu.py)."""
    logW = model.tindex.leaf_path_logweights()
    ptr, idx = rc["ptr"], rc["idx"]
    w = np.repeat(np.exp(logW), np.diff(ptr))
    s2 = np.where(var <= 0, EPS, var)                       # src/common.jl:137
    m = np.bincount(idx, weights=w * mu, minlength=n_t)
    m2 = np.bincount(idx, weights=w * mu * mu, minlength=n_t)
    sv = np.bincount(idx, weights=w * s2, minlength=n_t)
    if model.root.kind == "gp":
        return m, sv
    return m, sv + (m2 - m * m)


def _aggregate_dsmgp(model, xt, ptr, mu, var, sel_cache=None):
    """Sum/product aggregation of the leaf moments exactly as `_minpredict` + `_predict`
    (`src/common.jl:134-196,275-302`) do it, from ONE set of leaf predictions."""
    n_t = xt.shape[0]
    all_rows = np.arange(n_t, dtype=np.int64)
    sel_cache = {} if sel_cache is None else sel_cache

    def child_masks(node, rows):
        """getchild once per split node: the three recursions below partition the rows identically."""
        m = sel_cache.get(node.id)
        if m is None:
            ch = get_child(node, xt[rows])
            m = sel_cache[node.id] = [ch == k for k in range(len(node.children))]
        return m

    def leaf_vals(node):
        a, b = ptr[node.leaf], ptr[node.leaf + 1]
        return mu[a:b], var[a:b]

    def minpred(node, rows):
        if node.kind == "gp":
            return leaf_vals(node)[0]
        if node.kind == "split":
            out = np.zeros(rows.size)
            for sel, c in zip(child_masks(node, rows), node.children):
                out[sel] = minpred(c, rows[sel])
            return out
        out = np.full(rows.size, np.inf)
        for c in node.children:
            out = np.minimum(out, minpred(c, rows))
        return out

    def pred(node, rows, mmin):
        if node.kind == "gp":
            m, s2 = leaf_vals(node)
            s2 = np.where(s2 <= 0, EPS, s2)  # src/common.jl:137
            if not np.all(m >= mmin):
                raise AssertionError("leaf mean below the shift (src/common.jl:138)")
            with np.errstate(divide="ignore"):
                return np.log(m - mmin), np.log(m * m), np.log(s2)
        if node.kind == "split":
            lm = np.zeros(rows.size)
            lm2 = np.zeros(rows.size)
            ls = np.zeros(rows.size)
            for sel, c in zip(child_masks(node, rows), node.children):
                a, b, d = pred(c, rows[sel], mmin[sel])
                lm[sel], lm2[sel], ls[sel] = a, b, d
            return lm, lm2, ls
        K = len(node.children)
        lm = np.zeros((rows.size, K))
        lm2 = np.zeros((rows.size, K))
        ls = np.zeros((rows.size, K))
        for k, c in enumerate(node.children):
            a, b, d = pred(c, rows, mmin)
            lm[:, k] = a + node.logweights[k]
            lm2[:, k] = b + node.logweights[k]
            ls[:, k] = d + node.logweights[k]
        return _logsumexp(lm, axis=1), _logsumexp(lm2, axis=1), _logsumexp(ls, axis=1)

    def predict_node(node, rows):
        if node.kind == "gp":                    # src/common.jl:175-179
            m, s2 = leaf_vals(node)
            return m, np.where(s2 <= 0, EPS, s2)
        if node.kind == "split":                 # src/common.jl:243-254
            m = np.zeros(rows.size)
            v = np.zeros(rows.size)
            for sel, c in zip(child_masks(node, rows), node.children):
                m[sel], v[sel] = predict_node(c, rows[sel])
            return m, v
        mmin = minpred(node, rows)               # src/common.jl:294-302
        lm, lm2, ls = pred(node, rows, mmin - 1.0)
        m = np.exp(lm) + mmin - 1.0
        return m, np.exp(ls) + (np.exp(lm2) - m * m)

    return predict_node(model.root, all_rows)


def _prior_diag(lf, xt):
    k = lf.kernel
    if k.kind == KIND_ISO_SE:
        return np.full(xt.shape[0], np.exp(2 * k.logs))
    if k.kind == KIND_ARD_SE:
        return np.full(xt.shape[0], np.exp(2 * k.logs) * xt.shape[1])
    return np.sum(xt * xt, axis=1) / np.exp(k.logl) ** 2


def _aggregate_poe(model, xt, ptr, mu, var):
    """PoE / gPoE / rBCM combination rules (`src/common.jl:145-149,198-273`)."""
    n_t = xt.shape[0]

    def poe(node):
        if node.kind == "gp":
            a, b = ptr[node.leaf], ptr[node.leaf + 1]
            return mu[a:b], 1.0 / var[a:b]
        m = np.zeros(n_t)
        t = np.zeros(n_t)
        for c in node.children:
            m_, t_ = poe(c)
            t += t_
            m += t_ * m_
        return m / t, t

    root = model.root
    if root.kind == "gp":
        m, t = poe(root)
        return m, 1.0 / t
    if model.family == "poe":
        m, t = poe(root)
        return m, 1.0 / t
    if model.family == "gpoe":
        beta = 1.0 / len(root.children)
        m = np.zeros(n_t)
        t = np.zeros(n_t)
        for c in root.children:
            m_, t_ = poe(c)
            t += beta * t_
            m += beta * t_ * m_
        return m / t, 1.0 / t
    lf0 = get_leaves(root)[0]
    s = _prior_diag(lf0, xt) + np.exp(2 * lf0.logNoise)
    Cc = 1.0 / s
    m = np.zeros(n_t)
    for c in root.children:
        m_, t_ = poe(c)
        s_ = 1.0 / t_
        beta = 0.5 * (np.log(s) - np.log(s_))
        Cc = Cc + (beta * t_) - (beta / s)
        m = m + m_ * (beta * t_)
    return m / Cc, 1.0 / Cc


# ------------------------------------------------------------------------------------ scores

def mse(y_true, y_pred):
    return float(np.mean((np.asarray(y_true) - np.asarray(y_pred)) ** 2))


def mae(y_true, y_pred):
    return float(np.mean(np.abs(np.asarray(y_true) - np.asarray(y_pred))))


def sse(y_true, y_pred):
    """standard error of the squared error (`src/scorefunctions.jl:9`; Julia's std is the unbiased one)."""
    se = (np.asarray(y_true) - np.asarray(y_pred)) ** 2
    return float(np.std(se, ddof=1) / np.sqrt(se.size))


def sae(y_true, y_pred):
    """`src/scorefunctions.jl:14`"""
    ae = np.abs(np.asarray(y_true) - np.asarray(y_pred))
    return float(np.std(ae, ddof=1) / np.sqrt(ae.size))


def nlpd(y_true, mu, var):
    """`src/scorefunctions.jl:16`."""
    y_true, mu, var = map(np.asarray, (y_true, mu, var))
    return float(np.mean(0.5 * np.log(2 * np.pi * var) + 0.5 * (y_true - mu) ** 2 / var))
