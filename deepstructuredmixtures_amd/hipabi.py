"""ctypes binding of the C ABI in include/dsmgp_hip.h (the same symbols Julia would `ccall`).

The GP-expert path has no CPU fallback: `Context()` raises when the library or a GPU is missing.
"""
import ctypes as C
import os
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdsmgp_hip.so")

N_TIMINGS = 21
AGG_MIXTURE, AGG_POE, AGG_GPOE, AGG_RBCM = 0, 1, 2, 3     # include/dsmgp_hip.h DSMGP_AGG_*
OPT_ARD_LENGTHSCALE_GRADIENT = 1
OPT_FUSED_GRAM = 2
OPT_FUSED_STEPS = 3
OPT_DIAG_IN_UPDATE = 4
OPT_FIT_GRAPH = 5
OPT_LANES = 6
SCORE_NAMES = ("mse", "sse", "mae", "sae", "nlpd")


def agg_width(family, n_groups=0):
    """Number of partial-sum vectors (of length n_t) the aggregation of a family exchanges."""
    return 3 if family == AGG_MIXTURE else (2 * int(n_groups) if family == AGG_RBCM else 2)
TIMING_NAMES = ("gram", "chol_update", "chol_diag", "chol_trsm", "solve", "mll", "predict_gram",
                "predict_update", "predict_trsm", "predict_var", "gradients", "total_fit", "total_predict", "chol_reduce", "alpha",
                "grad_inverse", "grad_contraction", "grad_traces", "chol_fused", "chol_update_union", "chol_fused_union")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_ctx = C.c_void_p

# every symbol include/dsmgp_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "dsmgp_create": (C.c_int, [C.c_int32, C.POINTER(_ctx)]),
    "dsmgp_destroy": (C.c_int, [_ctx]),
    "dsmgp_last_error": (C.c_char_p, [_ctx]),
    "dsmgp_device_name": (C.c_int, [_ctx, C.c_char_p, C.c_int32]),
    "dsmgp_set_train": (C.c_int, [_ctx, _dp, _dp, C.c_int64, C.c_int32]),
    "dsmgp_set_leaves": (C.c_int, [_ctx, C.c_int32, _lp, _lp, _ip, _dp]),
    "dsmgp_set_sharing": (C.c_int, [_ctx, _ip, _ip, _lp]),
    "dsmgp_set_hyper": (C.c_int, [_ctx, C.c_int32, C.c_int32, _dp, C.c_int32]),
    "dsmgp_fit": (C.c_int, [_ctx, _dp, _ip, _dp]),
    "dsmgp_set_test": (C.c_int, [_ctx, _dp, C.c_int64, C.c_int32, _lp, _lp]),
    "dsmgp_set_tree": (C.c_int, [_ctx, C.c_int64, C.POINTER(C.c_int8), _lp, _lp, _lp, _dp, C.c_int64, _lp]),
    "dsmgp_set_test_routed": (C.c_int, [_ctx, _dp, C.c_int64, C.c_int32]),
    "dsmgp_routes": (C.c_int, [_ctx, _lp, _lp]),
    "dsmgp_predict_run": (C.c_int, [_ctx, _dp]),
    "dsmgp_predict_fetch": (C.c_int, [_ctx, _dp, _dp]),
    "dsmgp_predict_leaves": (C.c_int, [_ctx, _dp, C.c_int64, C.c_int32, _lp, _lp, _dp, _dp]),
    "dsmgp_gradients": (C.c_int, [_ctx, _dp, C.c_int32]),
    "dsmgp_set_gradient_leaves": (C.c_int, [_ctx, _ip]),
    "dsmgp_set_option": (C.c_int, [_ctx, C.c_int32, C.c_int32]),
    "dsmgp_lanes": (C.c_int, [_ctx, _ip]),
    "dsmgp_aggregate": (C.c_int, [_ctx, C.c_int32, _dp, _ip, C.c_int32, C.c_int32, C.c_int32, _dp, _dp]),
    "dsmgp_aggregate_partial": (C.c_int, [_ctx, C.c_int32, _dp, _ip, C.c_int32, _dp]),
    "dsmgp_aggregate_finish": (C.c_int, [_ctx, _dp, C.c_int32, C.c_int32, _dp, _dp]),
    "dsmgp_scores": (C.c_int, [_ctx, _dp, _dp]),
    "dsmgp_kernel_matrix": (C.c_int, [_ctx, C.c_int32, _dp, C.c_int64, _dp, C.c_int64, _dp]),
    "dsmgp_download_factor": (C.c_int, [_ctx, C.c_int32, _dp, _dp]),
    "dsmgp_set_profile": (C.c_int, [_ctx, C.c_int32]),
    "dsmgp_set_joint": (C.c_int, [_ctx, C.c_int32]),
    "dsmgp_timings": (C.c_int, [_ctx, _dp]),
    "dsmgp_work": (C.c_int, [_ctx, _dp, _ip]),
    "dsmgp_work_fused": (C.c_int, [_ctx, _dp, _ip]),
    "dsmgp_work_gradients": (C.c_int, [_ctx, _dp, _dp, _ip]),
    "dsmgp_release": (C.c_int, [_ctx]),
    "dsmgp_reserve": (C.c_int, [_ctx, C.c_int64]),
    "dsmgp_overlap_main": (C.c_int, [C.c_int32, _lp, _lp, C.c_int64, _lp, _lp]),
    "dsmgp_comm_unique_id": (C.c_int, [C.c_char_p]),
    "dsmgp_comm_init": (C.c_int, [_ctx, C.c_int32, C.c_int32, C.c_char_p]),
    "dsmgp_allgather": (C.c_int, [_ctx, _dp, C.c_int64, _dp]),
    "dsmgp_fit_exchange": (C.c_int, [_ctx, C.c_int64, _dp]),
    "dsmgp_aggregate_exchange": (C.c_int, [_ctx, _dp]),
    "dsmgp_aggregate_exchange_empty": (C.c_int, [_ctx, C.c_int32, C.c_int64, _dp]),
    "dsmgp_comm_destroy": (C.c_int, [_ctx]),
    "dsmgp_tree_build": (C.c_int, [_dp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                   C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "dsmgp_tree_sizes": (C.c_int, [C.c_void_p, _lp, _lp, _lp, _lp]),
    "dsmgp_tree_export": (C.c_int, [C.c_void_p, _ip, _ip, _ip, _dp, _dp, _lp, _dp, _lp, _lp, _dp]),
    "dsmgp_tree_means": (C.c_int, [C.c_void_p, _dp, C.c_int64, _dp]),
    "dsmgp_tree_free": (C.c_int, [C.c_void_p]),
    "dsmgp_tree_route": (C.c_int, [C.c_int64, C.POINTER(C.c_int8), _lp, _lp, _lp, _dp, C.c_int64, _lp, C.c_int64, _dp, C.c_int64,
                                   C.c_int64, C.c_int64, C.c_int64, _lp, _lp, C.c_int64, _lp]),
    "dsmgp_estimate_bytes": (C.c_int64, [C.c_int32, _lp, _lp, C.c_int32, C.c_int32]),
    "dsmgp_memory": (C.c_int, [_ctx, _lp, _lp]),
    "dsmgp_probe_f64_mfma": (C.c_int, [_ctx, _dp]),
    "dsmgp_probe_f64_mfma_detail": (C.c_int, [_ctx, C.c_int32, _dp]),
    "dsmgp_clock_sample_start": (C.c_int, [_ctx, C.c_double]),
    "dsmgp_clock_sample_read": (C.c_int, [_ctx, _dp, _dp]),
}

# include/dsmgp_hip_diag.h: only in libdsmgp_hip_diag.so (csrc/build.sh diag), used by tools/ for kernel tuning
DIAG_SIGNATURES = {
    "dsmgp_probe_coissue": (C.c_int, [_ctx, _dp]),
    "dsmgp_bench_tile": (C.c_int, [_ctx, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "dsmgp_probe_diag": (C.c_int, [_ctx, C.c_int32, C.c_int32, C.c_int32, _dp, _dp]),
    "dsmgp_bench_fused8": (C.c_int, [_ctx, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "dsmgp_probe_diag_fused": (C.c_int, [_ctx, C.c_int32, C.c_int32, C.c_int32, _dp]),
}
DIAG_LIB_PATH = os.path.join(_HERE, "libdsmgp_hip_diag.so")

_lib = None
_diag_lib = None


# include/dsmgp_hip.h DSMGP_E_*
E_ARG, E_STATE, E_HIP, E_NOMEM, E_NODEVICE, E_DOMAIN = -1, -2, -3, -4, -5, -6


class DsmgpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dsmgp error {code}: {msg}")
        self.code = code


class DsmgpDomainError(DsmgpError, ValueError):
    """DSMGP_E_DOMAIN: a test row outside the region of a split node -- the ValueError the host routing raises for the same row."""


def load_diag_library():
    """The diagnostic build (tools/ only): same ABI plus the entry points of include/dsmgp_hip_diag.h."""
    global _diag_lib
    if _diag_lib is None:
        if not os.path.exists(DIAG_LIB_PATH):
            raise RuntimeError(f"{DIAG_LIB_PATH} is missing: build it with deepstructuredmixtures_amd/csrc/build.sh diag")
        lib = C.CDLL(DIAG_LIB_PATH)
        for name, (res, args) in {**SIGNATURES, **DIAG_SIGNATURES}.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _diag_lib = lib
    return _diag_lib


def load_library():
    """dlopen libdsmgp_hip.so and attach prototypes; raises if the extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with deepstructuredmixtures_amd/csrc/build.sh "
                           "(there is no CPU fallback for the GP-expert path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _f64_fortran(a):
    a = np.asfortranarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_lp)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


class Context:
    """One GPU context (dsmgp_ctx)."""

    def __init__(self, device=0, diag=False):
        self.lib = load_diag_library() if diag else load_library()
        self.h = _ctx()
        rc = self.lib.dsmgp_create(int(device), C.byref(self.h))
        if rc != 0:
            msg = self.lib.dsmgp_last_error(None).decode()
            self.h = None
            raise DsmgpError(rc, msg)
        self.L = 0
        self.D = 0
        self.route_total = 0
        self.n_t = 0

    def _chk(self, rc):
        if rc != 0:
            raise (DsmgpDomainError if rc == E_DOMAIN else DsmgpError)(rc, self.lib.dsmgp_last_error(self.h).decode())

    def close(self):
        if self.h is not None:
            self.lib.dsmgp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._chk(self.lib.dsmgp_device_name(self.h, buf, 256))
        return buf.value.decode()

    def set_train(self, X, y):
        X, px = _f64_fortran(X)
        y, py = _f64(y)
        assert X.ndim == 2 and y.shape == (X.shape[0],)
        self._chk(self.lib.dsmgp_set_train(self.h, px, py, X.shape[0], X.shape[1]))
        self.D = X.shape[1]

    def set_leaves(self, obs_ptr, obs_idx, kernel_id, mean):
        obs_ptr, p0 = _i64(obs_ptr)
        obs_idx, p1 = _i64(obs_idx)
        kernel_id, p2 = _i32(kernel_id)
        mean, p3 = _f64(mean)
        L = len(obs_ptr) - 1
        self._chk(self.lib.dsmgp_set_leaves(self.h, L, p0, p1, p2, p3))
        self.L = L

    def set_sharing(self, op, src, plen):
        if op is None:
            self._chk(self.lib.dsmgp_set_sharing(self.h, None, None, None))
            return
        op, p0 = _i32(op)
        src, p1 = _i32(src)
        plen, p2 = _i64(plen)
        self._chk(self.lib.dsmgp_set_sharing(self.h, p0, p1, p2))

    def set_hyper(self, kernel_id, kind, loghyp):
        v, p = _f64(loghyp)
        self._chk(self.lib.dsmgp_set_hyper(self.h, int(kernel_id), int(kind), p, len(v)))

    def fit(self):
        """Returns (mll[L], info[L], device_seconds)."""
        mll = np.empty(self.L)
        info = np.empty(self.L, dtype=np.int32)
        sec = C.c_double(0.0)
        self._chk(self.lib.dsmgp_fit(self.h, mll.ctypes.data_as(_dp), info.ctypes.data_as(_ip), C.byref(sec)))
        return mll, info, sec.value

    def _test_matrix(self, Xt):
        """The test rows as the ABI reads them (n_t x D doubles, column-major); a matrix of another width is refused HERE,
        before any pointer is handed over (the library checks D once more: DSMGP_E_ARG)."""
        Xt, px = _f64_fortran(Xt)
        if Xt.ndim != 2 or Xt.shape[1] != self.D:
            raise ValueError(f"test matrix of shape {Xt.shape}: the model was trained on D = {self.D} columns")
        return Xt, px

    def set_test(self, Xt, route_ptr, route_idx):
        Xt, px = self._test_matrix(Xt)
        route_ptr, p0 = _i64(route_ptr)
        route_idx, p1 = _i64(route_idx)
        self._chk(self.lib.dsmgp_set_test(self.h, px, Xt.shape[0], Xt.shape[1], p0, p1))
        self.route_total = int(route_ptr[-1])
        self.n_t = int(Xt.shape[0])

    def set_tree(self, kind, first_child, n_child, split_dim, thr, leaf_local):
        """The model's tree as flat arrays (tree._RouteIndex) with `leaf_local` = index of every region in THIS context's leaf
        table (-1: held by another rank): what `set_test_routed` walks on the device."""
        kind = np.ascontiguousarray(kind, dtype=np.int8)
        first_child, p1 = _i64(first_child)
        n_child, p2 = _i64(n_child)
        split_dim, p3 = _i64(split_dim)
        thr = np.ascontiguousarray(thr, dtype=np.float64)
        leaf_local, p5 = _i64(leaf_local)
        self._chk(self.lib.dsmgp_set_tree(self.h, int(kind.size), kind.ctypes.data_as(C.POINTER(C.c_int8)), p1, p2, p3,
                                          thr.ctypes.data_as(_dp), int(thr.shape[1]), p5))

    def set_test_routed(self, Xt):
        """`set_test` with the routing of predict done on the device (needs `set_tree`)."""
        Xt, px = self._test_matrix(Xt)
        self._chk(self.lib.dsmgp_set_test_routed(self.h, px, Xt.shape[0], Xt.shape[1]))
        self.n_t = int(Xt.shape[0])
        ptr = np.zeros(self.L + 1, dtype=np.int64)
        self._chk(self.lib.dsmgp_routes(self.h, ptr.ctypes.data_as(_lp), None))
        self.route_total = int(ptr[-1])
        return ptr

    def routes(self):
        """(route_ptr, route_idx) of the registered test set."""
        ptr = np.zeros(self.L + 1, dtype=np.int64)
        self._chk(self.lib.dsmgp_routes(self.h, ptr.ctypes.data_as(_lp), None))
        idx = np.zeros(max(1, int(ptr[-1])), dtype=np.int64)
        self._chk(self.lib.dsmgp_routes(self.h, ptr.ctypes.data_as(_lp), idx.ctypes.data_as(_lp)))
        return ptr, idx[:int(ptr[-1])]

    def predict_run(self):
        sec = C.c_double(0.0)
        self._chk(self.lib.dsmgp_predict_run(self.h, C.byref(sec)))
        return sec.value

    def predict_fetch(self):
        mu = np.empty(self.route_total)
        var = np.empty(self.route_total)
        self._chk(self.lib.dsmgp_predict_fetch(self.h, mu.ctypes.data_as(_dp), var.ctypes.data_as(_dp)))
        return mu, var

    def predict_leaves(self, Xt, route_ptr, route_idx):
        self.set_test(Xt, route_ptr, route_idx)
        self.predict_run()
        return self.predict_fetch()

    def set_option(self, option, value):
        """include/dsmgp_hip.h DSMGP_OPT_*: OPT_ARD_LENGTHSCALE_GRADIENT = 1, OPT_FUSED_GRAM = 2, OPT_FUSED_STEPS = 3."""
        self._chk(self.lib.dsmgp_set_option(self.h, int(option), int(value)))

    def lanes(self):
        """Leaf lanes of the current plan (1 or 2; 0 before the first fit of a leaf table)."""
        n = C.c_int32(0)
        self._chk(self.lib.dsmgp_lanes(self.h, C.byref(n)))
        return n.value

    def set_gradient_leaves(self, active=None):
        """Restrict `gradients` to the leaves with a true flag (None: all): the other rows come back as zeros."""
        if active is None:
            self._chk(self.lib.dsmgp_set_gradient_leaves(self.h, None))
            return
        a = np.ascontiguousarray(np.asarray(active) != 0, dtype=np.int32)
        assert a.size == self.L
        self._chk(self.lib.dsmgp_set_gradient_leaves(self.h, a.ctypes.data_as(_ip)))

    def gradients(self, stride):
        g = np.zeros((self.L, stride))
        self._chk(self.lib.dsmgp_gradients(self.h, g.ctypes.data_as(_dp), int(stride)))
        return g

    # ---- predict(model, x) aggregation + scores on the device (src/common.jl:134-302, src/scorefunctions.jl) ----
    def _agg_args(self, family, leaf_coef, leaf_group):
        coef = pc = grp = pg = None
        if leaf_coef is not None:
            coef, pc = _f64(leaf_coef)
            assert coef.size == self.L
        if leaf_group is not None:
            grp, pg = _i32(leaf_group)
            assert grp.size == self.L
        return coef, pc, grp, pg

    def aggregate(self, family, leaf_coef=None, leaf_group=None, n_groups=0, plain=False, prior_kernel_id=0, fetch=True):
        """(mu, var) of length n_t from the moments of the last predict_run; fetch=False leaves them in HBM (scores)."""
        coef, pc, grp, pg = self._agg_args(family, leaf_coef, leaf_group)
        mu = np.empty(self.n_t) if fetch else None
        var = np.empty(self.n_t) if fetch else None
        self._chk(self.lib.dsmgp_aggregate(self.h, int(family), pc, pg, int(n_groups), 1 if plain else 0, int(prior_kernel_id),
                                           mu.ctypes.data_as(_dp) if fetch else None,
                                           var.ctypes.data_as(_dp) if fetch else None))
        return mu, var

    def aggregate_partial(self, family, leaf_coef=None, leaf_group=None, n_groups=0, fetch=True):
        """This context's partial sums, shape (W, n_t): what ranks / contexts exchange and add (fetch=False: they stay on
        the device, for aggregate_exchange)."""
        coef, pc, grp, pg = self._agg_args(family, leaf_coef, leaf_group)
        part = np.empty((agg_width(family, n_groups), self.n_t)) if fetch else None
        self._chk(self.lib.dsmgp_aggregate_partial(self.h, int(family), pc, pg, int(n_groups),
                                                   part.ctypes.data_as(_dp) if fetch else None))
        return part

    def aggregate_finish(self, partial=None, plain=False, prior_kernel_id=0, fetch=True):
        pp = None
        if partial is not None:
            partial, pp = _f64(partial)
        mu = np.empty(self.n_t) if fetch else None
        var = np.empty(self.n_t) if fetch else None
        self._chk(self.lib.dsmgp_aggregate_finish(self.h, pp, 1 if plain else 0, int(prior_kernel_id),
                                                  mu.ctypes.data_as(_dp) if fetch else None,
                                                  var.ctypes.data_as(_dp) if fetch else None))
        return mu, var

    def scores(self, y_test):
        """dict(mse, sse, mae, sae, nlpd) of the aggregated prediction still on the device."""
        y, py = _f64(y_test)
        assert y.size == self.n_t
        out = np.zeros(5)
        self._chk(self.lib.dsmgp_scores(self.h, py, out.ctypes.data_as(_dp)))
        return dict(zip(SCORE_NAMES, out.tolist()))

    # ---- exchange over RCCL (what a Julia host binds for the multi-GPU path; the Python mirror itself exchanges
    #      through torch.distributed, backend "nccl" = the same RCCL) ----
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        rc = load_library().dsmgp_comm_unique_id(buf)
        if rc != 0:
            raise DsmgpError(rc, load_library().dsmgp_last_error(None).decode())
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        assert len(unique_id) == 128
        self._chk(self.lib.dsmgp_comm_init(self.h, int(rank), int(world), unique_id))
        self.world = int(world)

    def allgather(self, local):
        local, p = _f64(local)
        out = np.empty((self.world, local.size))
        self._chk(self.lib.dsmgp_allgather(self.h, p, local.size, out.ctypes.data_as(_dp)))
        return out

    def fit_exchange(self, count):
        """(world, count, 2) array of (mll, info) per rank and leaf slot of the last fit, gathered device to device."""
        out = np.empty((self.world, int(count), 2))
        self._chk(self.lib.dsmgp_fit_exchange(self.h, int(count), out.ctypes.data_as(_dp)))
        return out

    def aggregate_exchange(self, W, fetch=False):
        """Sum over ranks of the partial sums of the last aggregate_partial, left on the device for aggregate_finish."""
        tot = np.empty((int(W), self.n_t)) if fetch else None
        self._chk(self.lib.dsmgp_aggregate_exchange(self.h, tot.ctypes.data_as(_dp) if fetch else None))
        return tot

    def aggregate_exchange_empty(self, W, n_t):
        """The same collective from a rank without leaves: contributes zeros, returns the total (W, n_t)."""
        tot = np.empty((int(W), int(n_t)))
        self._chk(self.lib.dsmgp_aggregate_exchange_empty(self.h, int(W), int(n_t), tot.ctypes.data_as(_dp)))
        return tot

    def comm_destroy(self):
        self._chk(self.lib.dsmgp_comm_destroy(self.h))

    def kernel_matrix(self, kernel_id, x1, x2):
        x1, p1 = _f64_fortran(x1)
        x2, p2 = _f64_fortran(x2)
        K = np.empty((x1.shape[0], x2.shape[0]), order="F")
        self._chk(self.lib.dsmgp_kernel_matrix(self.h, int(kernel_id), p1, x1.shape[0], p2, x2.shape[0],
                                               K.ctypes.data_as(_dp)))
        return K

    def download_factor(self, leaf, n, factor=True):
        """(gp.cK.factors lower triangle, gp.alpha) of one leaf; factor=False fetches alpha only (F is None)."""
        F = np.empty((n, n), order="F") if factor else None
        alpha = np.empty(n)
        self._chk(self.lib.dsmgp_download_factor(self.h, int(leaf), F.ctypes.data_as(_dp) if factor else None,
                                                 alpha.ctypes.data_as(_dp)))
        return F, alpha

    def set_joint(self, on):
        self._chk(self.lib.dsmgp_set_joint(self.h, 1 if on else 0))

    def release(self):
        """Free the factors and everything else that scales with the leaf sizes (streaming mode)."""
        self._chk(self.lib.dsmgp_release(self.h))

    def reserve(self, nbytes):
        """One device pool for the large arenas of all following leaf tables (0 drops it)."""
        self._chk(self.lib.dsmgp_reserve(self.h, int(nbytes)))

    def set_profile(self, level):
        """0/False: totals only; 1: update launches only; 2/True: every kernel category; 3: as 1 under the kernel names of 0."""
        lv = 2 if level is True else int(level)
        self._chk(self.lib.dsmgp_set_profile(self.h, lv))

    def timings(self):
        t = np.zeros(N_TIMINGS)
        self._chk(self.lib.dsmgp_timings(self.h, t.ctypes.data_as(_dp)))
        return dict(zip(TIMING_NAMES, t.tolist()))

    def work(self):
        f = C.c_double(0.0)
        n = C.c_int32(0)
        self._chk(self.lib.dsmgp_work(self.h, C.byref(f), C.byref(n)))
        return f.value, n.value

    def work_fused(self):
        """(algorithmic flops, launches) of the fused tile launches of the last fit (update + solve of fused block steps)."""
        f = C.c_double(0.0)
        n = C.c_int32(0)
        self._chk(self.lib.dsmgp_work_fused(self.h, C.byref(f), C.byref(n)))
        return f.value, n.value

    def work_gradients(self):
        """(algorithmic flops of L^-T, of the contraction, number of contraction tiles) of dsmgp_gradients."""
        a, b, n = C.c_double(0.0), C.c_double(0.0), C.c_int32(0)
        self._chk(self.lib.dsmgp_work_gradients(self.h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def memory(self):
        a = C.c_int64(0)
        b = C.c_int64(0)
        self._chk(self.lib.dsmgp_memory(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def probe_f64_mfma_detail(self, blocks_per_cu=4):
        """dict(tflops, cycles_per_mfma, clock_ghz, waves_per_simd) of a register-only f64 MFMA loop."""
        out = np.zeros(4)
        self._chk(self.lib.dsmgp_probe_f64_mfma_detail(self.h, int(blocks_per_cu), out.ctypes.data_as(_dp)))
        return dict(zip(("tflops", "cycles_per_mfma", "clock_ghz", "waves_per_simd"), out.tolist()))

    def clock_sample_start(self, milliseconds):
        """Start a shader-clock sample of `milliseconds` beside whatever this context launches next (returns at once)."""
        self._chk(self.lib.dsmgp_clock_sample_start(self.h, float(milliseconds)))

    def clock_sample_read(self):
        """(GHz held, milliseconds covered) of the sample started last; waits for it."""
        g, ms = C.c_double(0.0), C.c_double(0.0)
        self._chk(self.lib.dsmgp_clock_sample_read(self.h, C.byref(g), C.byref(ms)))
        return g.value, ms.value

    def probe_coissue(self):
        out = np.zeros(9)
        self._chk(self.lib.dsmgp_probe_coissue(self.h, out.ctypes.data_as(_dp)))
        return {m: dict(mfma_tflops=out[3 * i], valu_tflops=out[3 * i + 1], ms=out[3 * i + 2])
                for i, m in enumerate(("mfma_only", "valu_only", "both"))}

    def probe_diag(self, ntiles, ld=8192, reps=20):
        """(diagnostic library) microseconds per launch of the diagonal-block kernel on ntiles blocks, phases of one block"""
        us = C.c_double(0.0)
        ph = np.zeros(23)
        self._chk(self.lib.dsmgp_probe_diag(self.h, int(ntiles), int(ld), int(reps), C.byref(us), ph.ctypes.data_as(_dp)))
        return us.value, ph

    def probe_diag_fused(self, ntiles, K=0, reps=10):
        """(diagnostic library) microseconds per launch of the fused steps' diagonal-block task on ntiles synthetic blocks"""
        us = C.c_double(0.0)
        self._chk(self.lib.dsmgp_probe_diag_fused(self.h, int(ntiles), int(K), int(reps), C.byref(us)))
        return us.value

    def bench_tile(self, ntiles, K, mode=0, group=16, reps=3):
        """TFLOP/s of the tile GEMM on a uniform batch (diagnostic)."""
        s = C.c_double(0.0)
        self._chk(self.lib.dsmgp_bench_tile(self.h, int(ntiles), int(K), int(mode), int(group), int(reps), C.byref(s)))
        return 2.0 * 128 * 128 * K * ntiles / s.value / 1e12

    def bench_fused8(self, ntasks, K, group=16, reps=3):
        """seconds per launch of the eight-wave fused tile task on a uniform batch (diagnostic)."""
        s = C.c_double(0.0)
        self._chk(self.lib.dsmgp_bench_fused8(self.h, int(ntasks), int(K), int(group), int(reps), C.byref(s)))
        return s.value

    def probe_f64_mfma(self):
        t = C.c_double(0.0)
        self._chk(self.lib.dsmgp_probe_f64_mfma(self.h, C.byref(t)))
        return t.value


class MultiContext:
    """Several contexts on ONE GPU, each with a share of the leaves and its own HIP stream, driven concurrently
    from host threads (the ABI calls release the GIL).  While one context sits in the latency-bound panel phases
    of its factorisation (diagonal blocks, panel solves, split-K reduces), the other's update launches fill the
    chip.  Leaves are independent, so results are exactly those of the single-context run per leaf.  Used for the
    per-rank shard of multi-GPU runs, where launches are too small to fill the GPU on their own."""

    def __init__(self, device=0, n=2):
        from concurrent.futures import ThreadPoolExecutor
        self.subs = [Context(device) for _ in range(n)]
        self.pool = ThreadPoolExecutor(max_workers=n)
        self.L = 0
        self.route_total = 0
        self._leaves = None
        self._sharing = (None, None, None)
        self._dirty = True
        self._test = None
        self._test_dirty = False
        self.part = None
        self.act = self.subs          # contexts that hold leaves (all of them unless there are fewer leaf groups)

    def close(self):
        for s in self.subs:
            s.close()
        self.pool.shutdown(wait=False)

    def device_name(self):
        return self.subs[0].device_name()

    def _each(self, fn):
        return list(self.pool.map(fn, self.act))

    def set_train(self, X, y):
        for s in self.subs:
            s.set_train(X, y)
        self._dirty = True

    def set_leaves(self, obs_ptr, obs_idx, kernel_id, mean):
        self._leaves = (np.asarray(obs_ptr, dtype=np.int64), np.asarray(obs_idx, dtype=np.int64),
                        np.asarray(kernel_id, dtype=np.int32), np.asarray(mean, dtype=np.float64))
        self.L = len(obs_ptr) - 1
        self._sharing = (None, None, None)
        self._dirty = True
        self._test = None

    def set_sharing(self, op, src, plen):
        self._sharing = (None, None, None) if op is None else (np.asarray(op), np.asarray(src), np.asarray(plen))
        self._dirty = True

    def set_hyper(self, kernel_id, kind, loghyp):
        for s in self.subs:
            s.set_hyper(kernel_id, kind, loghyp)

    def set_joint(self, on):
        for s in self.subs:
            s.set_joint(on)

    def set_option(self, option, value):
        for s in self.subs:
            s.set_option(option, value)
        self._test_dirty = self._test is not None     # OPT_FUSED_GRAM drops a registered test set with the plan

    def set_profile(self, on):
        for s in self.subs:
            s.set_profile(on)

    def _partition(self):
        """Leaves -> sub-context by longest-processing-time on n^3, sharing groups kept together."""
        ptr = self._leaves[0]
        n = np.diff(ptr).astype(np.float64)
        op, src, _ = self._sharing
        group = np.arange(self.L)
        if op is not None:
            for j in range(self.L):
                if op[j] != 0 and src[j] >= 0:
                    group[j] = src[j]
        cost = np.zeros(self.L)
        for j in range(self.L):
            cost[group[j]] += n[j] ** 3 if (op is None or op[j] == 0) else n[j] ** 2
        load = np.zeros(len(self.subs))
        owner_of = {}
        for g in sorted(set(group.tolist()), key=lambda g: -cost[g]):
            r = int(np.argmin(load))
            owner_of[g] = r
            load[r] += cost[g]
        owner = np.array([owner_of[g] for g in group])
        return [np.flatnonzero(owner == r) for r in range(len(self.subs))]

    def _upload(self):
        if not self._dirty:
            return
        ptr, idx, kid, mean = self._leaves
        op, src, plen = self._sharing
        parts = self._partition()
        self.act = [s for s, loc in zip(self.subs, parts) if len(loc)]
        self.part = [loc for loc in parts if len(loc)]
        for s, loc in zip(self.act, self.part):
            lptr = np.concatenate([[0], np.cumsum(ptr[loc + 1] - ptr[loc])])
            lidx = np.concatenate([idx[ptr[g]:ptr[g + 1]] for g in loc]) if len(loc) else np.zeros(0, np.int64)
            s.set_leaves(lptr, lidx, kid[loc], mean[loc])
            if op is None:
                s.set_sharing(None, None, None)
            else:
                g2l = {int(g): i for i, g in enumerate(loc)}
                s.set_sharing(op[loc], np.array([g2l.get(int(src[g]), -1) for g in loc], dtype=np.int32), plen[loc])
        self._dirty = False
        self._test_dirty = self._test is not None

    def _upload_test(self):
        if not self._test_dirty:
            return
        Xt, rptr, ridx = self._test
        for s, loc in zip(self.act, self.part):
            lptr = np.concatenate([[0], np.cumsum(rptr[loc + 1] - rptr[loc])])
            lidx = np.concatenate([ridx[rptr[g]:rptr[g + 1]] for g in loc]) if len(loc) else np.zeros(0, np.int64)
            s.set_test(Xt, lptr, lidx)
        self._test_dirty = False

    def fit(self):
        self._upload()
        self._upload_test()
        res = self._each(lambda s: s.fit())
        mll = np.empty(self.L)
        info = np.empty(self.L, dtype=np.int32)
        for (m, i, _), loc in zip(res, self.part):
            mll[loc] = m
            info[loc] = i
        return mll, info, max(r[2] for r in res)

    def set_test(self, Xt, route_ptr, route_idx):
        self._test = (np.asfortranarray(Xt, dtype=np.float64), np.asarray(route_ptr, dtype=np.int64),
                      np.asarray(route_idx, dtype=np.int64))
        self.route_total = int(route_ptr[-1])
        self._test_dirty = True
        if not self._dirty:
            self._upload_test()

    def predict_run(self):
        self._upload()
        self._upload_test()
        return max(self._each(lambda s: s.predict_run()))

    def predict_fetch(self):
        _, rptr, _ = self._test
        mu = np.empty(self.route_total)
        var = np.empty(self.route_total)
        for s, loc in zip(self.act, self.part):
            m, v = s.predict_fetch()
            pos = 0
            for g in loc:
                c = int(rptr[g + 1] - rptr[g])
                mu[rptr[g]:rptr[g + 1]] = m[pos:pos + c]
                var[rptr[g]:rptr[g + 1]] = v[pos:pos + c]
                pos += c
        return mu, var

    def predict_leaves(self, Xt, route_ptr, route_idx):
        self.set_test(Xt, route_ptr, route_idx)
        self.predict_run()
        return self.predict_fetch()

    def aggregate_partial(self, family, leaf_coef=None, leaf_group=None, n_groups=0):
        """Partial sums of all sub-contexts added in context order (the sums are linear in the leaves)."""
        coef = None if leaf_coef is None else np.asarray(leaf_coef, dtype=np.float64)
        grp = None if leaf_group is None else np.asarray(leaf_group, dtype=np.int32)
        tot = None
        for s, loc in zip(self.act, self.part):
            p = s.aggregate_partial(family, None if coef is None else coef[loc], None if grp is None else grp[loc], n_groups)
            tot = p if tot is None else tot + p
        return tot

    def set_gradient_leaves(self, active=None):
        self._upload()
        a = None if active is None else np.asarray(active) != 0
        for s, loc in zip(self.act, self.part):
            s.set_gradient_leaves(None if a is None else a[loc])

    def gradients(self, stride):
        self._upload()
        res = self._each(lambda s: s.gradients(stride))
        g = np.zeros((self.L, stride))
        for r, loc in zip(res, self.part):
            g[loc] = r
        return g

    def timings(self):
        ts = [s.timings() for s in self.act]
        return {k: max(t[k] for t in ts) for k in ts[0]}

    def work(self):
        ws = [s.work() for s in self.act]
        return sum(w[0] for w in ws), max(w[1] for w in ws)

    def work_fused(self):
        ws = [s.work_fused() for s in self.act]
        return sum(w[0] for w in ws), max(w[1] for w in ws)

    def work_gradients(self):
        ws = [s.work_gradients() for s in self.act]
        return sum(w[0] for w in ws), sum(w[1] for w in ws), sum(w[2] for w in ws)

    def probe_f64_mfma(self):
        return self.subs[0].probe_f64_mfma()


def estimate_bytes(n, n_test, D, with_gradients=False):
    """Device bytes a group of leaves with sizes n (and n_test routed rows each) needs resident."""
    lib = load_library()
    n = np.ascontiguousarray(n, dtype=np.int64)
    nt = None if n_test is None else np.ascontiguousarray(n_test, dtype=np.int64)
    return int(lib.dsmgp_estimate_bytes(len(n), n.ctypes.data_as(_lp), None if nt is None else nt.ctypes.data_as(_lp),
                                        int(D), 1 if with_gradients else 0))


def overlap_main(obs_ptr, obs_idx, N):
    """(main, c_main) of the sharing schedule for a leaf table in CSR form (host routine of the library, no device):
    main[j] = argmax_i D[i,j] D[j,i], c_main[j] = |obs_j n obs_main[j]|."""
    lib = load_library()
    ptr = np.ascontiguousarray(obs_ptr, dtype=np.int64)
    idx = np.ascontiguousarray(obs_idx, dtype=np.int64)
    L = ptr.size - 1
    main = np.zeros(L, dtype=np.int64)
    cm = np.zeros(L, dtype=np.int64)
    rc = lib.dsmgp_overlap_main(L, ptr.ctypes.data_as(_lp), idx.ctypes.data_as(_lp), int(N), main.ctypes.data_as(_lp),
                                cm.ctypes.data_as(_lp))
    if rc != 0:
        raise DsmgpError(rc, "dsmgp_overlap_main: bad leaf table")
    return main, cm


def tree_route(kind, first_child, n_child, split_dim, thr, leaf_id, n_leaves, xt, reach):
    """CSR (route_ptr, route_idx) of the test rows every leaf predicts, from the flat tree arrays of tree._RouteIndex (host
    routine of the library, no device; include/dsmgp_hip.h dsmgp_tree_route).  reach: the most leaves one row can reach."""
    lib = load_library()
    x = np.asarray(xt, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    if not (x.flags.c_contiguous or x.flags.f_contiguous):
        x = np.ascontiguousarray(x)
    rs, cs = x.strides[0] // 8, x.strides[1] // 8
    ptr = np.zeros(int(n_leaves) + 1, dtype=np.int64)
    idx = np.empty(int(x.shape[0]) * int(reach), dtype=np.int64)
    nr = C.c_int64(0)
    rc = lib.dsmgp_tree_route(int(kind.size), kind.ctypes.data_as(C.POINTER(C.c_int8)), first_child.ctypes.data_as(_lp),
                              n_child.ctypes.data_as(_lp), split_dim.ctypes.data_as(_lp), thr.ctypes.data_as(_dp), int(thr.shape[1]),
                              leaf_id.ctypes.data_as(_lp), int(n_leaves), x.ctypes.data_as(_dp), int(x.shape[0]), int(x.shape[1]),
                              int(rs), int(cs),
                              ptr.ctypes.data_as(_lp), idx.ctypes.data_as(_lp), int(idx.size), C.byref(nr))
    if rc != 0:
        if rc == -1 and split_dim.size and x.shape[1] <= int(split_dim.max()):
            raise IndexError(f"test rows have {x.shape[1]} columns, the tree splits on dimension {int(split_dim.max())}")
        raise ValueError("test point outside the region of a split node (reference loops forever here)" if rc == E_DOMAIN
                         else f"dsmgp_tree_route failed ({rc}): malformed tree arrays")
    return ptr, idx[:nr.value].copy() if nr.value < idx.size else idx


def tree_build(X, min_data, n_splits, n_sum_children, depth, bnoise, sum_root, n_kernels, seed, y=None):
    """Node table of the random partition tree (host routine of the library, no device; include/dsmgp_hip.h
    dsmgp_tree_build): dict of arrays kind / parent / split_dim / lb / ub / thr_ptr / thr / obs_ptr / obs / dir_u, and,
    given y, `mean` = mean(y[obs]) per region (dsmgp_tree_means)."""
    lib = load_library()
    X, px = _f64_fortran(X)
    N, D = X.shape
    h = C.c_void_p()
    rc = lib.dsmgp_tree_build(px, N, D, int(min_data), int(n_splits), int(n_sum_children), int(depth), float(bnoise),
                              1 if sum_root else 0, int(n_kernels), C.c_uint64(int(seed)), C.byref(h))
    if rc != 0:
        raise DsmgpError(rc, "dsmgp_tree_build: bad arguments (non-finite inputs?)")
    try:
        n, nthr, nobs, ndir = (C.c_int64(0) for _ in range(4))
        lib.dsmgp_tree_sizes(h, C.byref(n), C.byref(nthr), C.byref(nobs), C.byref(ndir))
        n, nthr, nobs, ndir = n.value, nthr.value, nobs.value, ndir.value
        out = dict(kind=np.empty(n, np.int32), parent=np.empty(n, np.int32), split_dim=np.empty(n, np.int32),
                   lb=np.empty((n, D)), ub=np.empty((n, D)), thr_ptr=np.empty(n + 1, np.int64), thr=np.empty(max(nthr, 1)),
                   obs_ptr=np.empty(n + 1, np.int64), obs=np.empty(max(nobs, 1), np.int64), dir_u=np.empty(max(ndir, 1)))
        rc = lib.dsmgp_tree_export(h, out["kind"].ctypes.data_as(_ip), out["parent"].ctypes.data_as(_ip),
                                   out["split_dim"].ctypes.data_as(_ip), out["lb"].ctypes.data_as(_dp), out["ub"].ctypes.data_as(_dp),
                                   out["thr_ptr"].ctypes.data_as(_lp), out["thr"].ctypes.data_as(_dp),
                                   out["obs_ptr"].ctypes.data_as(_lp), out["obs"].ctypes.data_as(_lp), out["dir_u"].ctypes.data_as(_dp))
        if rc != 0:
            raise DsmgpError(rc, "dsmgp_tree_export failed")
        out["thr"], out["obs"], out["dir_u"] = out["thr"][:nthr], out["obs"][:nobs], out["dir_u"][:ndir]
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float64)
            if y.shape != (N,):
                raise ValueError("tree_build: y must have one value per row of X")
            out["mean"] = np.empty(max(int(np.count_nonzero(out["kind"] == 0)), 1))
            rc = lib.dsmgp_tree_means(h, y.ctypes.data_as(_dp), N, out["mean"].ctypes.data_as(_dp))
            if rc != 0:
                raise DsmgpError(rc, "dsmgp_tree_means failed")
        return out
    finally:
        lib.dsmgp_tree_free(h)


class StreamingContext:
    """Factor-and-discard mode for leaf tables that do not fit in HBM (config 5 at depth 2 needs 1.4 TB
    per kernel).  The leaves are packed into groups that fit a byte budget; one pass over the groups does, per
    group: upload -> fit (the resident test rows ride along) -> fetch mll / predictive moments [-> gradients] ->
    release.  Only per-leaf results survive a group (log-marginals, moments, gradient vectors): factors are
    recomputed by the next pass, which is what `train!` does anyway (every iteration changes the hyper-parameters).
    Interface of `Context`; results per leaf are those of the resident run."""

    def __init__(self, device=0, budget_bytes=None, headroom=0.85):
        self.ctx = Context(device)
        self.budget = budget_bytes
        self.headroom = headroom
        self.L = 0
        self.D = 0
        self.route_total = 0
        self._leaves = None
        self._sharing = (None, None, None)
        self._hyper = {}
        self._test = None
        self.want_gradients = 0        # stride of the gradient rows to collect during the pass (0 = none)
        self.keep_alpha = ()           # global leaf ids whose alpha is fetched before their group is released
        self.groups = None
        self._res = None               # results of the last pass
        self.passes = 0
        self._pool_bytes = 0
        self._auto_budget = None
        self.verbose = bool(os.environ.get("DSMGP_STREAM_VERBOSE"))   # one progress line per leaf group on stderr

    def close(self):
        self.ctx.close()

    def device_name(self):
        return self.ctx.device_name()

    def set_train(self, X, y):
        self.ctx.set_train(X, y)
        self.D = np.asarray(X).shape[1]
        self._res = None

    def set_leaves(self, obs_ptr, obs_idx, kernel_id, mean):
        self._leaves = (np.asarray(obs_ptr, dtype=np.int64), np.asarray(obs_idx, dtype=np.int64),
                        np.asarray(kernel_id, dtype=np.int32), np.asarray(mean, dtype=np.float64))
        self.L = len(obs_ptr) - 1
        self._sharing = (None, None, None)
        self._test = None
        self.groups = None
        self._res = None

    def set_sharing(self, op, src, plen):
        self._sharing = (None, None, None) if op is None else (np.asarray(op), np.asarray(src), np.asarray(plen))
        self.groups = None
        self._res = None

    def set_hyper(self, kernel_id, kind, loghyp):
        self.ctx.set_hyper(kernel_id, kind, loghyp)
        self._res = None

    def set_joint(self, on):
        pass                           # the test rows always ride along here: factors do not outlive their group

    def set_option(self, option, value):
        self.ctx.set_option(option, value)   # every group registers its leaves and test rows again anyway
        self._res = None

    def set_profile(self, on):
        self.ctx.set_profile(on)

    def set_test(self, Xt, route_ptr, route_idx):
        self._test = (np.asfortranarray(Xt, dtype=np.float64), np.asarray(route_ptr, dtype=np.int64),
                      np.asarray(route_idx, dtype=np.int64))
        self.route_total = int(route_ptr[-1])
        self.groups = None
        self._res = None

    def _make_groups(self):
        ptr = self._leaves[0]
        n = np.diff(ptr)
        nt = np.diff(self._test[1]) if self._test is not None else np.zeros(self.L, dtype=np.int64)
        op, src, _ = self._sharing
        unit = np.arange(self.L)
        if op is not None:
            for j in range(self.L):
                if op[j] != 0 and src[j] >= 0:
                    unit[j] = src[j]           # leaves sharing a factor travel together
        budget = self.budget
        if budget is None:
            if self._pool_bytes:               # the pool holds the memory: keep the budget it was sized for
                budget = self._auto_budget
            else:
                budget = self._auto_budget = int(self.ctx.memory()[1] * self.headroom)
        units = {}
        for j in range(self.L):
            units.setdefault(int(unit[j]), []).append(j)
        sized = []
        for u, members in units.items():
            m = np.array(members)
            # COPY leaves alias the factor of their source: count the factor once per unit
            own = m if op is None else m[op[m] != 1]
            b = estimate_bytes(n[own], nt[own], self.D, bool(self.want_gradients))
            if op is not None and np.any(op[m] == 1):
                cp = m[op[m] == 1]
                b += int(np.sum((nt[cp] + 128) * (n[cp] + 128) * 8))
            sized.append((b, members))
        sized.sort(key=lambda t: -t[0])
        if sized and sized[0][0] > budget:
            raise DsmgpError(-4, f"one leaf group needs {sized[0][0] >> 20} MiB, budget is {budget >> 20} MiB")
        groups, loads = [], []
        for b, members in sized:                   # first-fit decreasing
            for gi in range(len(groups)):
                if loads[gi] + b <= budget:
                    groups[gi].extend(members)
                    loads[gi] += b
                    break
            else:
                groups.append(list(members))
                loads.append(b)
        self.groups = [np.array(sorted(g)) for g in groups]
        # one pool for all groups (several of them: the arenas would otherwise be allocated and freed per group,
        # and the driver clears memory on allocation); sized for the fullest group plus the split-K workspaces
        want = 0 if len(self.groups) < 2 else int(max(loads)) + (3 << 30)
        if want != self._pool_bytes:
            self.ctx.reserve(want)
            self._pool_bytes = want

    def _pass(self):
        """One sweep over the groups; fills self._res."""
        if self.groups is None:
            self._make_groups()
        ptr, idx, kid, mean = self._leaves
        op, src, plen = self._sharing
        mll = np.empty(self.L)
        info = np.zeros(self.L, dtype=np.int32)
        mu = np.empty(self.route_total)
        var = np.empty(self.route_total)
        grads = np.zeros((self.L, self.want_gradients)) if self.want_gradients else None
        alphas = {}
        seconds, tpred = 0.0, 0.0
        tsum = {}
        flops, launches = 0.0, 0
        fflops, flaunches = 0.0, 0
        import time as _time
        host = {"set_leaves": 0.0, "set_sharing": 0.0, "set_test": 0.0, "fit": 0.0, "release": 0.0}   # wall seconds

        def timed(key, fn, *a):
            t0 = _time.perf_counter()
            r = fn(*a)
            host[key] += _time.perf_counter() - t0
            return r

        t_pass = _time.perf_counter()
        for gi, loc in enumerate(self.groups):
            if self.verbose:
                print(f"# streaming pass {self.passes + 1}: group {gi + 1}/{len(self.groups)} ({len(loc)} leaves) at "
                      f"{_time.perf_counter() - t_pass:.1f} s", file=sys.stderr, flush=True)
            c = self.ctx
            lptr = np.concatenate([[0], np.cumsum(ptr[loc + 1] - ptr[loc])])
            timed("set_leaves", c.set_leaves, lptr, np.concatenate([idx[ptr[g]:ptr[g + 1]] for g in loc]), kid[loc], mean[loc])
            if op is None:
                timed("set_sharing", c.set_sharing, None, None, None)
            else:
                g2l = {int(g): i for i, g in enumerate(loc)}
                timed("set_sharing", c.set_sharing, op[loc],
                      np.array([g2l.get(int(src[g]), -1) for g in loc], dtype=np.int32), plen[loc])
            if self._test is not None:
                Xt, rptr, ridx = self._test
                tptr = np.concatenate([[0], np.cumsum(rptr[loc + 1] - rptr[loc])])
                timed("set_test", c.set_test, Xt, tptr, np.concatenate([ridx[rptr[g]:rptr[g + 1]] for g in loc]))
            m, i, s = timed("fit", c.fit)
            mll[loc], info[loc] = m, i
            seconds += s
            if self._test is not None:
                tpred += c.predict_run()
                gm, gv = c.predict_fetch()
                pos = 0
                for g in loc:
                    k = int(rptr[g + 1] - rptr[g])
                    mu[rptr[g]:rptr[g + 1]] = gm[pos:pos + k]
                    var[rptr[g]:rptr[g + 1]] = gv[pos:pos + k]
                    pos += k
            if self.want_gradients:
                grads[loc] = c.gradients(self.want_gradients)
            for g in self.keep_alpha:
                li = np.flatnonzero(loc == g)
                if li.size:
                    alphas[int(g)] = c.download_factor(int(li[0]), int(ptr[g + 1] - ptr[g]), factor=False)[1]
            for k_, v_ in c.timings().items():
                tsum[k_] = tsum.get(k_, 0.0) + v_
            f_, n_ = c.work()
            flops += f_
            launches += n_
            ff_, fn_ = c.work_fused()
            fflops += ff_
            flaunches += fn_
            timed("release", c.release)
        self.passes += 1
        self.host_seconds = host
        self._res = dict(mll=mll, info=info, mu=mu, var=var, grads=grads, seconds=seconds, tpred=tpred, timings=tsum,
                         work=(flops, launches), work_fused=(fflops, flaunches), has_test=self._test is not None, alpha=alphas)

    def fit(self):
        self._pass()
        r = self._res
        return r["mll"], r["info"], r["seconds"]

    def predict_run(self):
        if self._res is None or not self._res["has_test"]:
            self._pass()
        return self._res["tpred"]

    def predict_fetch(self):
        return self._res["mu"], self._res["var"]

    def predict_leaves(self, Xt, route_ptr, route_idx):
        self.set_test(Xt, route_ptr, route_idx)
        self.predict_run()
        return self.predict_fetch()

    def set_gradient_leaves(self, active=None):
        """Streaming passes compute the gradients of every leaf of a group while its factors are resident: no mask."""
        return None

    def gradients(self, stride):
        if self._res is None or self._res["grads"] is None or self._res["grads"].shape[1] < stride:
            self.want_gradients = int(stride)
            self.groups = None              # L^-1 doubles the footprint of a group
            self._pass()
        return self._res["grads"][:, :stride].copy()

    def timings(self):
        return dict(self._res["timings"]) if self._res else {k: 0.0 for k in TIMING_NAMES}

    def work(self):
        return self._res["work"] if self._res else (0.0, 0)

    def work_fused(self):
        return self._res["work_fused"] if self._res else (0.0, 0)

    def alpha(self, leaf):
        """alpha of a leaf listed in `keep_alpha` before the last pass (factors themselves are discarded)."""
        return self._res["alpha"][int(leaf)]

    def memory(self):
        return self.ctx.memory()
