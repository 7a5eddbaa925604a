"""Host-side sum-product tree: node types, random builder, leaf-overlap matrix, test-point routing.

This is the caller side of the hot path: its output -- the leaf table
(`obs` CSR, kernel ids, per-leaf means), the sharing schedule and the test-row routes -- is
what crosses the C ABI.  Behaviour follows the reference's builder
(`src/treeStructure.jl:4-307,405-437`), overlap (`src/fit.jl:12-39`), scheduler decisions
(`src/fit.jl:71-122,208-292`) and routing (`src/common.jl:101-122`); indices are 0-based here.
Random draws come from `datagen.Stream` (Julia's RNG stream cannot be reproduced).
"""
import gc
import itertools
import numpy as np

try:                                   # at import, not inside the first buildDSMGP (0.15 s of its first call otherwise)
    import scipy.sparse as sp
except ImportError:                    # pragma: no cover
    sp = None

from .datagen import Stream
from .kernels import ConstMean, KernelFunction

_ids = itertools.count(1)


def _gensym(tag):
    return f"{tag}#{next(_ids)}"


class GPNode:
    """Leaf expert (`src/DeepStructuredMixtures.jl:61-71`). `leaf` = position in getLeaves order."""
    kind = "gp"

    def __init__(self, obs, lb, ub, kernel, kernelid, mean, logNoise):
        self.id = _gensym("GP")
        self.obs = np.asarray(obs, dtype=np.int64)  # ascending original row indices, 0-based
        self.nobs = int(self.obs.size)
        self.lb = lb
        self.ub = ub
        self.kernel = kernel
        self.kernelid = int(kernelid)  # 0-based
        self.mean = mean
        self.logNoise = float(logNoise)
        self.dnoise = 0.0
        self.leaf = -1
        self.children = []


class GPSplitNode:
    """Product node over axis-aligned regions (`src/DeepStructuredMixtures.jl:52-59`)."""
    kind = "split"

    def __init__(self, lowerBound, upperBound, split):
        self.id = _gensym("split")
        self.children = []
        self.lowerBound = lowerBound
        self.upperBound = upperBound
        self.split = split  # list of (dim, threshold); the last threshold is upperBound[dim]


class GPSumNode:
    """Mixture node (`src/DeepStructuredMixtures.jl:40-45`); `of_gps` marks GPSumNode{GPNode}."""
    kind = "sum"

    def __init__(self, of_gps=False):
        self.id = _gensym("sum")
        self.children = []
        self._lw = np.zeros(0)
        self.of_gps = of_gps

    # Assigning a fresh array is counted, so that a level-order index holding views of the weights
    # (model.TreeIndex) knows when to re-attach them; in-place writes need no bookkeeping.
    assignments = 0

    @property
    def logweights(self):
        return self._lw

    @logweights.setter
    def logweights(self, v):
        self._lw = np.asarray(v, dtype=np.float64)
        GPSumNode.assignments += 1

    def add(self, child, logw):
        self.children.append(child)
        self.logweights = np.append(self._lw, logw)


class DSMGPConfig:
    """`src/DeepStructuredMixtures.jl:91-101`; K = splits per split node, V = children per sum node."""

    def __init__(self, meanFun, kernels, observationNoise, minData, K, V, depth, bnoise, sumRoot):
        self.meanFun = meanFun
        self.kernels = kernels
        self.observationNoise = float(observationNoise)
        self.minData = int(minData)
        self.K = int(K)
        self.V = int(V)
        self.depth = int(depth)
        self.bnoise = float(bnoise)
        self.sumRoot = bool(sumRoot)


def get_leaves(node):
    known = node.__dict__.get("_leaves") if node.kind != "gp" else None
    if known is not None:              # the root of a tree from the native builder: its regions in creation order
        return list(known)
    """Leaves in depth-first child order (`src/fit.jl:9-10`)."""
    if node.kind == "gp":
        return [node]
    out = []
    for c in node.children:
        out.extend(get_leaves(c))
    return out


def ordered_nodes(node):
    out = []
    for c in node.children:
        out.extend(ordered_nodes(c))
    out.append(node)
    return out


# ----------------------------------------------------------------------------- builder

def _get_splits(xd, lower, upper, minData, eps, K, rng, depth=1):
    """Cut positions on one dimension (`src/treeStructure.jl:23-129`).

    `xd` is the column of the current region; `lower`/`upper` the bounds on that dimension.
    K_ starts at depth^2, so K=4 gives 3 cuts and K=8 gives 7.
    """
    K_ = depth * depth
    s = []
    l = max(lower, float(xd.min()))
    u = min(upper, float(xd.max()))
    v = u - l
    sel = xd[(xd > l) & (xd <= u)]
    if sel.size > 2 * minData:
        m = float(np.median(sel)) + 0.0     # -0.0 -> +0.0: a column holding both zeros has no unique median bit pattern
        z1 = z2 = 0
        c = 0
        s_new = m
        while z1 == 0 or z2 == 0:
            a = rng.beta22() * v + l
            s_new = float(eps * a + (1.0 - eps) * m)
            z1 = int(np.count_nonzero(sel <= s_new))
            z2 = sel.size - z1
            c += 1
            if c > 100:
                return s
        first_low = rng.randint(1, 2) == 1
        order = ("low", "high") if first_low else ("high", "low")
        for pos, side in enumerate(order):
            z = z1 if side == "low" else z2
            if z > minData and K_ < K:
                if side == "low":
                    s.extend(_get_splits(xd, lower, s_new, minData, eps, K, rng, depth + 1))
                else:
                    s.extend(_get_splits(xd, s_new, upper, minData, eps, K, rng, depth + 1))
                if pos == 0:
                    K_ += 1
        s.append(s_new)
    return s


def _build_gp(X, y, lb, ub, config, observations):
    """`src/treeStructure.jl:245-307`."""
    ym = float(np.mean(y)) if y.size else 0.0
    mfun = ConstMean(ym) if config.meanFun is None else config.meanFun
    if isinstance(config.kernels, (list, tuple)):
        w = config._rng.dirichlet1(len(config.kernels))
        node = GPSumNode(of_gps=True)
        for v, kern in enumerate(config.kernels):
            node.add(GPNode(observations, lb, ub, kern.copy(), v, mfun, config.observationNoise), np.log(w[v]))
        return node
    return GPNode(observations, lb, ub, config.kernels.copy(), 0, mfun, config.observationNoise)


def _build_split(X, y, lowerBound, upperBound, config, depth, observations, d=0):
    """`src/treeStructure.jl:131-210`. Region membership is (lb, ub] on dimension d."""
    rng = config._rng
    xd = X[:, d]
    s = sorted(_get_splits(xd, lowerBound[d], upperBound[d], config.minData, config.bnoise, config.K, rng))
    if not s:
        idx = np.flatnonzero((xd > lowerBound[d]) & (xd <= upperBound[d]))
        return _build_gp(X[idx], y[idx], lowerBound.copy(), upperBound.copy(), config, observations[idx])
    split = [(d, si) for si in s] + [(d, float(upperBound[d]))]
    node = GPSplitNode(lowerBound, upperBound, split)
    lb = lowerBound.copy()
    ub = upperBound.copy()
    for (_, si) in split:
        lb_ = lb.copy()
        ub_ = ub.copy()
        ub_[d] = si
        idx = np.flatnonzero((xd > lb_[d]) & (xd <= ub_[d]))
        if depth < config.depth and idx.size > config.minData:
            if config.sumRoot:
                child = _build_sum(X[idx], y[idx], lb_, ub_, config, depth, observations[idx])
            else:
                child = _build_split(X[idx], y[idx], lb_, ub_, config, depth, observations[idx])
        else:
            child = _build_gp(X[idx], y[idx], lb_, ub_, config, observations[idx])
        node.children.append(child)
        lb[d] = si
    return node


def _build_sum(X, y, lowerBound, upperBound, config, depth, observations):
    """`src/treeStructure.jl:212-243`: V children, each split on a dimension drawn ~ data range."""
    V = config.V
    node = GPSumNode()
    phi = X.max(axis=0) - X.min(axis=0)
    phi = phi / phi.sum() if phi.sum() > 0 else np.full(X.shape[1], 1.0 / X.shape[1])
    for _ in range(V):
        d = config._rng.categorical(phi)
        node.add(_build_split(X, y, lowerBound, upperBound, config, depth + 1, observations, d=d), -np.log(V))
    return node


def build_tree(X, y, config, seed=7, native=True):
    """`src/treeStructure.jl:4-21`.  native: the recursion runs in the library's host routine `dsmgp_tree_build`
    (one native pass over index lists; 26k nodes at depth 4 take 0.2 s instead of 3.3 s) and the node objects are made
    from its table; native=False is the interpreted builder below, kept as its line-by-line counterpart -- both draw
    from the same counter stream in the same order and return the same tree bit for bit (tests/test_host_cpu.py)."""
    if native:
        return _build_tree_native(X, y, config, seed)
    return build_tree_python(X, y, config, seed)


def _build_tree_native(X, y, config, seed):
    from . import hipabi
    N, D = X.shape
    assert N == y.shape[0] and np.all(np.isfinite(X))
    kvec = isinstance(config.kernels, (list, tuple))
    nk = len(config.kernels) if kvec else 0
    tab = hipabi.tree_build(X, config.minData, config.K, config.V, config.depth, config.bnoise, config.sumRoot, nk, seed,
                            y=y if config.meanFun is None else None)
    kind = tab["kind"].tolist()
    n = len(kind)
    nodes = [None] * n
    leaves = []
    sum_lw = np.full(config.V, -np.log(config.V))        # what V calls of GPSumNode.add(child, -log V) build up
    # tens of thousands of new containers and none to free: the cyclic collector would walk the growing tree again and
    # again (a third of this loop at depth 4), so it rests until the loop is through
    collect = gc.isenabled()
    gc.disable()
    try:
        _nodes_from_table(config, tab, nodes, leaves, kvec, nk)
    finally:
        if collect:
            gc.enable()
    for i in range(n):
        if kind[i] == 2:                                  # every split call under a sum node returns exactly one child
            assert len(nodes[i].children) == config.V
            nodes[i].logweights = sum_lw.copy()
    root = nodes[0]
    if root.kind != "gp":
        root._leaves = leaves        # creation order of the regions = get_leaves order (children are appended in creation order)
    return root


def _nodes_from_table(config, tab, nodes, leaves, kvec, nk):
    """Node objects of the native builder's table, in creation (pre-)order: parents come first."""
    kind, parent, sdim = tab["kind"].tolist(), tab["parent"].tolist(), tab["split_dim"].tolist()
    tptr, optr = tab["thr_ptr"].tolist(), tab["obs_ptr"].tolist()
    lbs, ubs, thr, obs_all = tab["lb"], tab["ub"], tab["thr"].tolist(), tab["obs"]
    means = tab["mean"].tolist() if config.meanFun is None else None
    noise = float(config.observationNoise)
    new, gp_cls, mean_cls = object.__new__, GPNode, ConstMean
    region = 0
    for i in range(len(kind)):
        k = kind[i]
        if k == 1:
            d = sdim[i]
            node = GPSplitNode(lbs[i], ubs[i], [(d, t) for t in thr[tptr[i]:tptr[i + 1]]])
        elif k == 2:
            node = GPSumNode()
        else:
            o0, o1 = optr[i], optr[i + 1]
            obs = obs_all[o0:o1]                          # views of the builder's tables: one allocation for all regions
            if means is None:
                mfun = config.meanFun
            else:
                mfun = new(mean_cls)
                mfun.m = means[region] if o1 > o0 else 0.0
            if kvec:
                u = tab["dir_u"][region * nk:(region + 1) * nk]
                e = -np.log(1.0 - u)                     # Stream.dirichlet1 on the uniforms the builder drew
                w = e / e.sum()
                node = GPSumNode(of_gps=True)
                for v, kern in enumerate(config.kernels):
                    node.add(GPNode(obs, lbs[i], ubs[i], kern.copy(), v, mfun, config.observationNoise), np.log(w[v]))
                for ch in node.children:
                    ch.leaf = len(leaves)
                    leaves.append(ch)
            else:
                # GPNode(obs, lb, ub, kernel.copy(), 0, mean, noise) without the conversions of its constructor: 18k
                # regions at depth 4, and the table already holds int64 rows and Python floats
                node = new(gp_cls)
                node.__dict__.update(id=_gensym("GP"), obs=obs, nobs=o1 - o0, lb=lbs[i], ub=ubs[i], kernel=config.kernels.copy(),
                                     kernelid=0, mean=mfun, logNoise=noise, dnoise=0.0, leaf=len(leaves), children=[])
                leaves.append(node)
            region += 1
        nodes[i] = node
        par = parent[i]
        if par >= 0:
            nodes[par].children.append(node)


def build_tree_python(X, y, config, seed=7):
    """The interpreted builder (`src/treeStructure.jl:4-21` and the recursions above)."""
    N, D = X.shape
    assert N == y.shape[0] and np.all(np.isfinite(X))
    config._rng = Stream(seed)
    lb = np.full(D, -np.inf)
    ub = np.full(D, np.inf)
    obs = np.arange(N, dtype=np.int64)
    if config.sumRoot:
        root = _build_sum(X, y, lb, ub, config, 0, obs)
    else:
        root = _build_split(X, y, lb, ub, config, 0, obs)
    for i, leaf in enumerate(get_leaves(root)):
        leaf.leaf = i
    return root


# ----------------------------------------------------------------------------- overlap + schedule

def obs_table(leaves):
    """CSR (ptr, idx) of the leaves' observation lists.  Leaves of the native builder hold consecutive views of one table
    (hipabi.tree_build): that table is returned as it is, without a copy; any other leaf list is concatenated."""
    L = len(leaves)
    ptr = np.zeros(L + 1, dtype=np.int64)
    if L:
        np.cumsum([lf.nobs for lf in leaves], out=ptr[1:])
    if L == 0:
        return ptr, np.zeros(0, np.int64)
    base = leaves[0].obs.base
    if (isinstance(base, np.ndarray) and base.ndim == 1 and base.dtype == np.int64 and base.flags.c_contiguous
            and base.size >= ptr[-1]):
        addr = base.__array_interface__["data"][0]
        start = leaves[0].obs.__array_interface__["data"][0] - addr
        if start >= 0 and start % 8 == 0 and start // 8 + ptr[-1] <= base.size and all(
                lf.obs.base is base and lf.obs.__array_interface__["data"][0] == addr + start + 8 * int(p)
                for lf, p in zip(leaves, ptr[:-1])):
            return ptr, base[start // 8:start // 8 + int(ptr[-1])]
    return ptr, np.concatenate([lf.obs for lf in leaves])


def _membership(leaves):
    """Sparse leaf-membership matrix M (L x N, int32 ones) and the leaf sizes."""
    if sp is None:
        raise ImportError("the leaf-overlap matrix needs scipy.sparse")
    L = len(leaves)
    nobs = np.array([lf.nobs for lf in leaves], dtype=np.int64)
    rows = np.repeat(np.arange(L), nobs)
    cols = obs_table(leaves)[1]
    N = int(cols.max()) + 1 if cols.size else 1
    return sp.csr_matrix((np.ones(rows.size, dtype=np.int32), (rows, cols)), shape=(L, N)), nobs


class LeafOverlap:
    """The leaf-overlap matrix of `src/fit.jl:12-39` kept as the sparse intersection counts C = M M^T
    (one kernel id; see get_overlap).  D[n,m] = 1 - (|n| - C[n,m]) / |n| where C > 0 and n != m, else 0 --
    at depth 4 the headline model has 18k leaves and 5.6e7 overlapping pairs: the dense L x L matrix of the
    reference and its O(L^2) bitset loops are what SURVEY 8(f).1 asks to remove."""

    def __init__(self, leaves):
        self._leaves = leaves
        self.nobs = np.array([lf.nobs for lf in leaves], dtype=np.int64)
        self.shape = (len(leaves), len(leaves))
        self._C = None
        self._pairs = None

    @property
    def C(self):
        """Sparse intersection counts M M^T (built on demand: the schedule itself uses the library's inverted-index
        routine and never needs them)."""
        if self._C is None:
            M, _ = _membership(self._leaves)
            C = (M @ M.T).tocsr()
            C.sort_indices()
            self._C = C
        return self._C

    def row(self, j):
        """Row j of the dense matrix: D[j, m] for every leaf m."""
        C = self.C
        out = np.zeros(self.shape[0])
        lo, hi = C.indptr[j], C.indptr[j + 1]
        cols, c = C.indices[lo:hi], C.data[lo:hi].astype(np.float64)
        nj = float(self.nobs[j])
        out[cols] = 1.0 - (nj - c) / nj
        out[j] = 0.0
        return out

    def todense(self):
        C = np.asarray(self.C.todense(), dtype=np.float64)
        n = self.nobs
        Dm = np.where(C > 0, 1.0 - (n[:, None] - C) / n[:, None], 0.0)
        np.fill_diagonal(Dm, 0.0)
        return Dm

    def main_pairs(self, native=True):
        """For every leaf j: main[j] = argmax_i D[i,j] D[j,i] (first maximum, 0 when the leaf overlaps nothing, like
        argmax of the dense all-zero column) and the two factors D[main,j], D[j,main]."""
        if self._pairs is not None:
            return self._pairs
        if native:
            # host routine of libdsmgp_hip.so (inverted index, one leaf at a time; include/dsmgp_hip.h)
            from . import hipabi
            ptr, idx = obs_table(self._leaves)
            main, c = hipabi.overlap_main(ptr, idx, int(idx.max()) + 1)
            n = self.nobs.astype(np.float64)
            cf = c.astype(np.float64)
            has = c > 0
            d_ij = np.where(has, 1.0 - (n[main] - cf) / n[main], 0.0)      # D[main, j]
            d_ji = np.where(has, 1.0 - (n - cf) / n, 0.0)                  # D[j, main]
            self._pairs = (main, d_ij, d_ji)
            return self._pairs
        C, n = self.C, self.nobs.astype(np.float64)
        L = self.shape[0]
        indptr, col = C.indptr, C.indices
        c = C.data.astype(np.float64)
        row = np.repeat(np.arange(L), np.diff(indptr))
        off = col != row
        d_rc = 1.0 - (n[row] - c) / n[row]          # D[row, col]
        d_cr = 1.0 - (n[col] - c) / n[col]          # D[col, row]
        prod = np.where(off, d_cr * d_rc, 0.0)
        rowmax = np.maximum.reduceat(prod, indptr[:-1])        # every row holds its diagonal entry
        pos = np.where((prod == rowmax[row]) & off, np.arange(col.size), col.size)
        first = np.minimum.reduceat(pos, indptr[:-1])
        has = (rowmax > 0.0) & (first < col.size)
        firstc = np.where(has, first, 0)
        main = np.where(has, col[firstc], 0).astype(np.int64)
        d_ij = np.where(has, d_cr[firstc], 0.0)                # D[main, j]
        d_ji = np.where(has, d_rc[firstc], 0.0)                # D[j, main]
        # a leaf without overlap points at leaf 0: D[0, j] and D[j, 0] are 0 unless j shares data with leaf 0,
        # in which case it has overlap and does not get here
        self._pairs = (main, d_ij, d_ji)
        return self._pairs


def get_overlap(root, L, sparse_from=1024):
    """Leaf-overlap matrix (`src/fit.jl:12-39`): D[n,m] = 1 - |n \\ m| / |n| for leaves under different
    children of a common sum node, forced to 1 when kernel ids differ.

    Two leaves that share an observation always hang under different children of some sum node (split nodes
    partition), and leaves that share none get 1 - |n|/|n| = 0, the default.  With a single kernel id the whole
    matrix therefore follows from the intersection counts C = M M^T of the sparse leaf-membership matrix M
    (L x N) -- one sparse product instead of the reference's O(L^2) bitset loops; beyond 1024 leaves (depth 3: 1,728 leaves, where the
    product and the dense L x L evaluation took 0.15 s of a 0.23 s build; `sparse_from`) the result
    stays sparse (LeafOverlap).  Kernel vectors (ids differ, forced ones also for disjoint pairs) take the
    literal pairwise recursion."""
    leaves = get_leaves(root)
    if len({lf.kernelid for lf in leaves}) > 1:
        return _get_overlap_pairwise(root, L)
    if L > sparse_from:
        return LeafOverlap(leaves)
    M, nobs = _membership(leaves)
    C = np.asarray((M @ M.T).todense(), dtype=np.float64)
    Dm = np.where(C > 0, 1.0 - (nobs[:, None] - C) / nobs[:, None], 0.0)   # same expression as src/fit.jl:30
    np.fill_diagonal(Dm, 0.0)
    return Dm


def _get_overlap_pairwise(root, L):
    Dm = np.zeros((L, L))

    def rec(node):
        if node.kind == "gp":
            return [node]
        if node.kind == "split":
            out = []
            for c in node.children:
                out.extend(rec(c))
            return out
        r = [rec(c) for c in node.children]
        for i in range(len(r)):
            for j in range(i + 1, len(r)):
                for nn in r[i]:
                    for mm in r[j]:
                        if nn.kernelid == mm.kernelid:
                            common = np.intersect1d(nn.obs, mm.obs, assume_unique=True).size
                            dn = nn.nobs - common
                            dm = mm.nobs - common
                        else:
                            dn = dm = 0
                        Dm[nn.leaf, mm.leaf] = 1.0 - dn / nn.nobs
                        Dm[mm.leaf, nn.leaf] = 1.0 - dm / mm.nobs
        out = []
        for x in r:
            out.extend(x)
        return out

    rec(root)
    return Dm


SHARE_FULL, SHARE_COPY, SHARE_PREFIX = 0, 1, 2


BRANCH_FULL, BRANCH_COPY, BRANCH_PREFIX, BRANCH_LOWRANK, BRANCH_LEADING = 0, 1, 2, 3, 4
BRANCH_NAMES = ("full", "copy", "prefix", "lowrank_as_full", "leading_as_full")


def share_decisions(leaves, Dm, tau=0.05):
    """Per-leaf factorisation decision of the shared-Cholesky `fit!` (`src/fit.jl:71-122`).

    Returns (op, src, plen, branch).  op[j] in {FULL, COPY, PREFIX} is what THIS implementation does: for COPY the
    factor of leaf src[j] is reused as is (`src/fit.jl:132-143`); for PREFIX the leading plen[j] x plen[j] block is
    leaf src[j]'s factor and the factorisation continues from column plen[j]
    (`src/fit.jl:208-292` -> `src/AdvancedCholeskey.jl:152-174`).  branch[j] (BRANCH_*) is the arm the REFERENCE takes
    for the leaf: besides full / copy / prefix,
      LOWRANK : its row-deletion branch with rows to delete (`src/fit.jl:174-201`: the leaf's list is a subset of its
                main leaf's and |main.obs[1:e] \ obs| / |obs| < tau).  Numerically defective there; a full
                factorisation here -- the leaves whose result differs from the reference's own
      LEADING : the same branch with nothing to delete (the list is a leading part of the main leaf's list; the
                reference takes the leading block of that factor, which is exact); a full factorisation here
    tau=0 disables sharing except COPY, as in the reference (`src/fit.jl:173,256`: 0 < 0 is false).
    """
    L = len(leaves)
    op = np.zeros(L, dtype=np.int32)
    src = np.full(L, -1, dtype=np.int32)
    plen = np.zeros(L, dtype=np.int64)
    branch = np.zeros(L, dtype=np.int8)
    if L == 0:
        return op, src, plen, branch
    if isinstance(Dm, LeafOverlap):
        main, d_ij, d_ji = Dm.main_pairs()
    else:
        main = np.empty(L, dtype=np.int64)
        for j in range(L):
            main[j] = int(np.argmax(Dm[:, j] * Dm[j, :]))
        d_ij = Dm[main, np.arange(L)]
        d_ji = Dm[np.arange(L), main]
    counts = np.bincount(main, minlength=L)
    order = sorted(range(L), key=lambda j: counts[j])  # stable, like Julia's sort! on objects
    processed = np.zeros(L, dtype=bool)
    for j in order:
        if processed[j]:
            continue
        i = int(main[j])
        if not processed[i]:
            processed[i] = True  # main leaf: full factorisation (`src/fit.jl:97-100`)
        processed[j] = True
        if i == j:
            continue
        lj, li = leaves[j], leaves[i]
        if li.kernelid != lj.kernelid or lj.obs[0] < li.obs[0]:
            continue
        ione = d_ij[j] == 1.0
        jone = d_ji[j] == 1.0
        if ione and jone:
            branch[j] = BRANCH_COPY
            if op[i] == SHARE_FULL:
                op[j], src[j] = SHARE_COPY, i
            elif op[i] == SHARE_COPY:
                op[j], src[j] = SHARE_COPY, src[i]
        elif ione and not jone and tau > 0.0:
            p = li.nobs
            if lj.obs[0] == li.obs[0] and lj.nobs > p and np.array_equal(lj.obs[:p], li.obs):
                branch[j] = BRANCH_PREFIX
                if op[i] == SHARE_FULL:
                    op[j], src[j], plen[j] = SHARE_PREFIX, i, p
        elif jone and not ione:
            e = int(np.searchsorted(li.obs, lj.obs[-1])) + 1       # position of max(obs) in the main leaf's list (:168)
            ndel = e - lj.nobs                                     # |main.obs[1:e] \ obs| (:170): obs is a subset of it
            if ndel / lj.nobs < tau:                               # :173
                branch[j] = BRANCH_LOWRANK if ndel > 0 else BRANCH_LEADING
    return op, src, plen, branch


def share_schedule(leaves, Dm, tau=0.05):
    """(op, src, plen) of `share_decisions`: the schedule `dsmgp_set_sharing` takes."""
    return share_decisions(leaves, Dm, tau)[:3]


def share_census(branch):
    """Counts of the reference's `fit!` arms over the leaves + the leaf ids of the two arms computed in full here
    (`share_decisions`): dict(full, copy, prefix, lowrank_as_full, leading_as_full, lowrank_leaves, leading_leaves)."""
    branch = np.asarray(branch)
    out = {name: int(np.count_nonzero(branch == code)) for code, name in enumerate(BRANCH_NAMES)}
    out["lowrank_leaves"] = np.flatnonzero(branch == BRANCH_LOWRANK).tolist()
    out["leading_leaves"] = np.flatnonzero(branch == BRANCH_LEADING).tolist()
    return out


# ----------------------------------------------------------------------------- routing

def get_child(node, x):
    """Child index per row (`src/common.jl:101-122`): first k with s_{k-1} < x[d] <= s_k."""
    d = node.split[0][0]
    th = np.array([s for (_, s) in node.split])
    idx = np.searchsorted(th, x[:, d], side="left")
    if np.any(idx >= len(th)):
        raise ValueError("test point outside the region of a split node (reference loops forever here)")
    return idx


class _RouteIndex:
    """The tree as flat arrays in breadth-first order (the children of a node are consecutive): what the library's host
    routine `dsmgp_tree_route` walks.  At depth 4 (5.6k split nodes, 18k leaves, 810k routed rows) the recursion over node
    objects took 0.09-0.18 s per test matrix, four times the prediction sweep it feeds."""

    def __init__(self, root):
        nodes = [root]
        kind, first, nchild, sdim, leaf, thr = [], [], [], [], [], []
        i = 0
        while i < len(nodes):
            nd = nodes[i]
            kind.append(0 if nd.kind == "gp" else 1 if nd.kind == "split" else 2)        # as dsmgp_tree_export numbers them
            first.append(len(nodes))
            nchild.append(len(nd.children))
            leaf.append(nd.leaf if nd.kind == "gp" else -1)
            sdim.append(nd.split[0][0] if nd.kind == "split" else 0)
            thr.append([t for (_, t) in nd.split] if nd.kind == "split" else [])
            nodes.extend(nd.children)
            i += 1
        self.kind = np.array(kind, dtype=np.int8)
        self.first = np.array(first, dtype=np.int64)
        self.nchild = np.array(nchild, dtype=np.int64)
        self.sdim = np.array(sdim, dtype=np.int64)
        self.leaf = np.array(leaf, dtype=np.int64)
        self.n_leaves = len(get_leaves(root))       # once: walking 26k node objects took 9 ms of every call at depth 4
        self.thr = np.full((len(nodes), max(1, max(len(t) for t in thr))), np.inf)
        for j, t in enumerate(thr):
            self.thr[j, :len(t)] = t
        reach = [1] * len(nodes)        # the most leaves one row can reach below a node: all children of a sum node, one of a split node
        for j in range(len(nodes) - 1, -1, -1):
            if kind[j]:
                below = reach[first[j]:first[j] + nchild[j]]
                reach[j] = sum(below) if kind[j] == 2 else max(below)
        self.reach = reach[0]


def route_index(root):
    """The flat arrays of the tree (`_RouteIndex`), built on first use and kept on the root."""
    ri = getattr(root, "_route_index", None)
    if ri is None:
        ri = root._route_index = _RouteIndex(root)
    return ri


def route(root, xt):
    """Which test rows each leaf is asked to predict: CSR (route_ptr, route_idx) in leaf order.

    Sum nodes forward every row to every child, split nodes to exactly one child
    (`src/common.jl:181-196,275-292`).  Walked by the library's host routine on the flat arrays of the tree (`_RouteIndex`,
    built on first use); `route_recursive` is the literal recursion it is tested against."""
    from . import hipabi
    ri = route_index(root)
    return hipabi.tree_route(ri.kind, ri.first, ri.nchild, ri.sdim, ri.thr, ri.leaf, ri.n_leaves, xt, ri.reach)


def route_recursive(root, xt):
    """`route` as the reference walks it: one recursion per node (`src/common.jl:181-196,275-292`)."""
    leaves = get_leaves(root)
    rows = [None] * len(leaves)

    def rec(node, idx):
        if node.kind == "gp":
            rows[node.leaf] = idx
        elif node.kind == "sum":
            for c in node.children:
                rec(c, idx)
        else:
            ch = get_child(node, xt[idx])
            for k, c in enumerate(node.children):
                rec(c, idx[ch == k])

    rec(root, np.arange(xt.shape[0], dtype=np.int64))
    ptr = np.zeros(len(leaves) + 1, dtype=np.int64)
    for i, r in enumerate(rows):
        ptr[i + 1] = ptr[i] + (0 if r is None else r.size)
    flat = np.concatenate([r for r in rows if r is not None]) if len(leaves) else np.zeros(0, np.int64)
    return ptr, flat.astype(np.int64)


def route_all(root, n_t):
    """PoE-family routing: every expert predicts every test row (`src/common.jl:198-208`)."""
    L = len(get_leaves(root))
    ptr = np.arange(L + 1, dtype=np.int64) * n_t
    return ptr, np.tile(np.arange(n_t, dtype=np.int64), L)
