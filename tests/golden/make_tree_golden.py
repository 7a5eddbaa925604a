"""tests/golden/tree_tables.npz: node tables of the random tree builder as oracle/tree.py (the literal restatement of
src/treeStructure.jl:4-307, drawing from the counter stream of SURVEY 8(d)) produces them, for BASELINE config 1 at full size (README
1-D sinusoid shape: N = 100, K = 4 splits, V = 3 sum children, M = 10, depth 2), a ragged PoE tree (no sum nodes: dimension 1 only,
src/treeStructure.jl:190) and a kernel-vector tree (Dirichlet weights).  Both product builders (csrc/host_tree.cpp, tree.py) must
reproduce them bit for bit: a fixture is what stays when all three implementations change together.

    python tests/golden/make_tree_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import tree as otree  # noqa: E402
from deepstructuredmixtures_amd.datagen import regression_data  # noqa: E402

CASES = {   # name: (N, D, data seed, M, splits, sum children, depth, eps, sum root, kernels, tree seed)
    "config1": (100, 1, 20201, 10, 4, 3, 2, 0.5, True, 0, 11),
    "poe": (1500, 3, 20203, 40, 8, 1, 2, 0.0, False, 0, 20203),
    "kvec": (1200, 4, 20205, 30, 4, 3, 2, 0.5, True, 2, 20205),
}

if __name__ == "__main__":
    out = {}
    for name, (N, D, dseed, M, K, V, depth, eps, sr, nk, seed) in CASES.items():
        X, y, _ = regression_data(N, D, seed=dseed)
        t = otree.table(otree.build_tree(X, y, M, K, V, depth, eps, sr, n_kernels=nk, seed=seed))
        for k in ("kind", "parent", "split_dim", "lb", "ub", "thr_ptr", "thr", "obs_ptr", "obs"):
            out[f"{name}/{k}"] = t[k]
        out[f"{name}/mean"] = np.asarray(t["mean"], dtype=np.float64)
        if nk:
            out[f"{name}/weights"] = np.asarray(t["weights"], dtype=np.float64)
        print(name, "nodes", t["kind"].size, "regions", int((t["kind"] == 0).sum()))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tree_tables.npz"), **out)
