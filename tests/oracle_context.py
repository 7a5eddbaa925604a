"""Test double for hipabi.Context backed by the CPU oracle.

Lets the CPU suite exercise the product's HOST logic (tree, sharing schedule, routing, aggregation,
sharding) end to end without a GPU.  Test infrastructure only -- never shipped, never timed."""
import numpy as np

from oracle import gp as ogp


class OracleContext:
    def __init__(self):
        self.hyper = {}
        self.gps = []
        self.L = 0

    def set_train(self, X, y):
        self.X = np.asarray(X, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)

    def set_leaves(self, obs_ptr, obs_idx, kernel_id, mean):
        self.obs = [np.asarray(obs_idx[obs_ptr[i]:obs_ptr[i + 1]]) for i in range(len(obs_ptr) - 1)]
        self.kid = list(kernel_id)
        self.mean = list(mean)
        self.L = len(self.obs)
        self.op = None

    def set_joint(self, on):
        pass

    def set_sharing(self, op, src, plen):
        self.op, self.src, self.plen = op, src, plen

    def set_hyper(self, kernel_id, kind, loghyp):
        self.hyper[int(kernel_id)] = (int(kind), np.array(loghyp, dtype=np.float64))

    def fit(self):
        self.gps = []
        for i in range(self.L):
            kind, h = self.hyper[self.kid[i]]
            k = ogp.make_kernel(kind, h[:-1])
            g = ogp.GaussianProcess(self.X[self.obs[i]], self.y[self.obs[i]], self.mean[i], k, h[-1], exact_dist=True)
            g.update_cholesky()
            self.gps.append(g)
        mll = np.array([g.mll() for g in self.gps])
        info = np.array([g.info for g in self.gps], dtype=np.int32)
        return mll, info, 0.0

    def set_test(self, Xt, route_ptr, route_idx):
        self.Xt = np.asarray(Xt, dtype=np.float64)
        self.rptr = np.asarray(route_ptr)
        self.ridx = np.asarray(route_idx)

    def predict_run(self):
        mu, var = [], []
        for i, g in enumerate(self.gps):
            rows = self.ridx[self.rptr[i]:self.rptr[i + 1]]
            if rows.size:
                m, v = g.prediction(self.Xt[rows])
                mu.append(m)
                var.append(v)
        self._mu = np.concatenate(mu) if mu else np.zeros(0)
        self._var = np.concatenate(var) if var else np.zeros(0)
        return 0.0

    def predict_fetch(self):
        return self._mu, self._var

    def predict_leaves(self, Xt, route_ptr, route_idx):
        self.set_test(Xt, route_ptr, route_idx)
        self.predict_run()
        return self.predict_fetch()

    def set_gradient_leaves(self, active=None):
        self._grad_active = None if active is None else np.asarray(active) != 0

    def gradients(self, stride):
        g = np.zeros((self.L, stride))
        act = getattr(self, "_grad_active", None)
        for i, gp_ in enumerate(self.gps):
            if act is not None and not act[i]:
                continue                           # not asked for: zeros, and no work (the count below is what a test reads)
            v = gp_.grad()
            g[i, : v.size] = v
            self.grad_evaluations = getattr(self, "grad_evaluations", 0) + 1
        return g


class OraclePartialContext(OracleContext):
    """OracleContext that also answers `aggregate_partial` (a NumPy restatement of agg_partial_kernel's sums), so the
    CPU suite drives the product's partial-sum exchange + host finish (`model._predict_device`) without a GPU."""

    def aggregate_partial(self, family, leaf_coef=None, leaf_group=None, n_groups=0):
        n_t = self.Xt.shape[0]
        leaf = np.repeat(np.arange(self.L), np.diff(self.rptr))
        mu, var, rows = self._mu, self._var, self.ridx
        if family == 0:
            w = np.asarray(leaf_coef)[leaf]
            v = np.where(var <= 0, 1e-8, var)
            return np.stack([np.bincount(rows, weights=w * mu, minlength=n_t),
                             np.bincount(rows, weights=w * mu * mu, minlength=n_t),
                             np.bincount(rows, weights=w * v, minlength=n_t)])
        t = 1.0 / var
        if family == 3:
            g = np.asarray(leaf_group)[leaf]
            out = np.zeros((2 * n_groups, n_t))
            for k in range(n_groups):
                sel = g == k
                out[2 * k] = np.bincount(rows[sel], weights=(t * mu)[sel], minlength=n_t)
                out[2 * k + 1] = np.bincount(rows[sel], weights=t[sel], minlength=n_t)
            return out
        b = np.asarray(leaf_coef)[leaf]
        return np.stack([np.bincount(rows, weights=b * t * mu, minlength=n_t), np.bincount(rows, weights=b * t, minlength=n_t)])
