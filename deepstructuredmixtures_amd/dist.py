"""Leaf sharding across ranks (one process per GPU) and the only exchange step of the path.

Leaves are independent units (`src/fit.jl:88-119` touches one leaf and at most its "main" leaf), so
each rank factorises and predicts its own leaves; what is exchanged is small: per-leaf log-marginals
after fit and the partial sums of the predict aggregation, by an all-gather.  With the `nccl` process group the
collective runs on the context's own stream through the library (`dsmgp_fit_exchange` / `dsmgp_aggregate_exchange`:
RCCL over xGMI, the payload never leaves HBM before it: `Shard.device_comm`); with gloo (CPU rehearsal in the
tests) or when that path cannot be set up, through `torch.distributed` from host buffers.
"""
import numpy as np


# The process group the exchanges of the path travel over: None = the default group.  A launcher may keep a host-side (gloo)
# default group for agreement and bookkeeping and hand the data path a second group of its own -- bench.py does: its RCCL group
# is a subgroup created and PROVEN beside a gloo default group, so that no rank ever depends on a half-built communicator and
# nothing has to be destroyed and re-initialised.
GROUP = None


def _pg():
    try:
        import torch.distributed as td
    except Exception:
        return None
    return td if td.is_available() and td.is_initialized() else None


def _backend(td):
    return td.get_backend(GROUP) if GROUP is not None else td.get_backend()


class Shard:
    """Owner rank of every leaf + gather helpers."""

    def __init__(self, owner, rank, world):
        self.owner = np.asarray(owner, dtype=np.int64)
        self.rank = int(rank)
        self.world = int(world)
        self.local = np.flatnonzero(self.owner == self.rank)
        self.comm_ctx = None          # hipabi.Context whose RCCL communicator carries the exchanges (device_comm)
        self.exchange_seconds = 0.0   # wall time this rank has spent inside the exchange collectives (waiting for the slowest
        self.exchanges = 0            # rank included): bench.py prints it per rank, so an N-GPU line explains its own efficiency
        self.exchange = "none" if self.world == 1 else "torch"
        self.standby_ctx = None       # a proven communicator not (yet) carrying the exchanges: device_comm(standby=True)
        self.rccl_ranks_seen = 0      # ranks whose value came back in the proof all-gather of the communicator

    # ---- exchange through the library (RCCL on the context's stream) ----------------------------
    @staticmethod
    def device_exchange_requested():
        """The in-library RCCL exchange is OPT-IN (`DSMGP_EXCHANGE=rccl-device` in the environment of every rank, read
        before the first GPU call): no run with more than one rank has exercised it on hardware yet (no multi-GPU box has
        been available to any round), so `torch.distributed` -- the path every world-2 test covers -- is the default."""
        import os
        return os.environ.get("DSMGP_EXCHANGE", "torch").strip().lower() == "rccl-device"

    def _all_min(self, ok):
        """MIN over ranks of a per-rank verdict, on whatever device the process group moves (gloo: host, nccl: GPU)."""
        import torch
        td = _pg()
        dev = torch.device("cuda", torch.cuda.current_device()) if _backend(td) == "nccl" else torch.device("cpu")
        flag = torch.tensor([int(ok)], dtype=torch.int32, device=dev)
        td.all_reduce(flag, op=td.ReduceOp.MIN, group=GROUP)
        return int(flag.item())

    def device_comm(self, ctx, force=False, standby=False):
        """Set up the library's own RCCL communicator on `ctx` (one per rank): rank 0 draws the 128-byte id,
        torch.distributed carries it to the others, every rank calls dsmgp_comm_init and proves the communicator with one
        tiny all-gather.  Taken when asked for (`device_exchange_requested`) under the `nccl` process group, or with
        `force` (tests: a one-rank communicator on the GPU box, stub contexts over gloo).

        Every rank runs the SAME torch.distributed collectives in the SAME order whatever fails where: (1) each rank
        probes that it can reach RCCL at all (draws an id of its own and drops it) and the verdicts are MIN-reduced --
        nobody enters the blocking dsmgp_comm_init unless everybody can; (2) rank 0's id is broadcast; (3) init, verdicts
        MIN-reduced; (4) the proof all-gather, verdicts MIN-reduced: all ranks take the device path or none does, and on any
        failure the exchanges stay on torch.distributed.  Returns the path in use.

        standby: build and prove the communicator but leave the exchanges on torch.distributed until
        `activate_device_exchange` (bench.py --gpus N: the communicator exists before the first timed series, which runs over
        torch.distributed; a second series of the same steps then runs over it -- one run, both paths)."""
        td = _pg()
        if self.world == 1 and force:          # one-rank communicator: the GPU box's test of this path
            uid = ctx.comm_unique_id()
            ctx.comm_init(0, 1, uid)
            self.comm_ctx, self.exchange = ctx, "rccl-device"
            return self.exchange
        if self.world == 1 or td is None or self.comm_ctx is not None or self.standby_ctx is not None:
            return self.exchange
        if not force and not (_backend(td) == "nccl" and self.device_exchange_requested()):
            return self.exchange
        uid = None
        try:
            uid = ctx.comm_unique_id()          # loads librccl: the probe of THIS rank (only rank 0's id is used)
        except Exception:
            uid = None
        if self._all_min(uid is not None) != 1:
            return self.exchange
        box = [uid if self.rank == 0 else None]
        td.broadcast_object_list(box, src=0)            # (the default group: object collectives are host work)
        ok = 1
        try:
            ctx.comm_init(self.rank, self.world, box[0])
        except Exception:
            ok = 0
        ok = self._all_min(ok)                 # nobody enters the proof collective unless every rank holds a communicator
        if ok == 1:
            try:
                got = np.asarray(ctx.allgather(np.array([float(self.rank)]))).ravel()
                self.rccl_ranks_seen = int(np.unique(got[(got >= 0) & (got < self.world)]).size)
                ok = int(np.array_equal(got, np.arange(self.world, dtype=np.float64)))
            except Exception:
                ok = 0
            ok = self._all_min(ok)
        if ok == 1 and standby:
            self.standby_ctx = ctx
        elif ok == 1:
            self.comm_ctx, self.exchange = ctx, "rccl-device"
            self.verified = False
        else:
            try:
                ctx.comm_destroy()
            except Exception:
                pass
        return self.exchange

    def activate_device_exchange(self):
        """The exchanges move onto the communicator `device_comm(standby=True)` left proven (every rank calls this at the same
        point of its program, or none: it is local, the collectives that follow are not).  The first exchange is cross-checked
        against torch.distributed as after a direct set-up (`fit_exchange`).  False: there is no standby communicator."""
        if self.standby_ctx is None:
            return False
        self.comm_ctx, self.standby_ctx, self.exchange = self.standby_ctx, None, "rccl-device"
        self.verified = False
        return True

    def drop_device_comm(self):
        """Back to torch.distributed for good (a cross-check of the device exchange failed)."""
        try:
            self.comm_ctx.comm_destroy()
        except Exception:
            pass
        self.comm_ctx, self.exchange = None, "torch"

    def fit_exchange(self, ctx, local_cols=None):
        """(mll, info) of every leaf after this rank's fit: one device-to-device all-gather (dsmgp_fit_exchange) of
        `count` = max leaves per rank slots of (mll, info) per rank; rank r's leaves, in leaf order, fill its first slots.
        With more than one rank the FIRST exchange of a communicator is cross-checked against the torch.distributed
        gather of `local_cols` (this rank's (n_local, 2) results); a mismatch on any rank drops the device path on all."""
        import time
        counts = np.bincount(self.owner, minlength=self.world)
        t0 = time.perf_counter()
        both = np.asarray(ctx.fit_exchange(int(max(1, counts.max()))))
        self.exchange_seconds += time.perf_counter() - t0
        self.exchanges += 1
        out = np.empty((self.owner.size, 2))
        for r in range(self.world):
            idx = np.flatnonzero(self.owner == r)
            out[idx] = both[r, : idx.size]
        if self.world > 1 and not getattr(self, "verified", True) and local_cols is not None:
            ref = self.gather_leaf_columns(local_cols)
            same = int(np.array_equal(ref, out))
            if self._all_min(same) != 1:
                self.drop_device_comm()
                return ref
            self.verified = True
        return out

    @staticmethod
    def single(L):
        return Shard(np.zeros(L, dtype=np.int64), 0, 1)

    # time model of one rank: flops at the sustained rate of the tile kernel + a serial term per block step of its
    # largest leaf (diagonal block + panel solve + split-K reduce + launch gaps sit on the chain of every step)
    RATE_FLOPS = 58e12
    STEP_SECONDS = 120e-6

    @staticmethod
    def lpt(nobs, op, src, rank, world, n_test=None):
        """Longest-processing-time greedy on cost n^3/3 + n^2 * n_t,leaf, where a rank's load also counts the
        block steps of its largest leaf (the batched factorisation runs max_leaf(n/128) dependent steps, so the
        rank holding the largest leaf gets a little less arithmetic); COPY/PREFIX leaves follow their source so
        shared factors stay on one GPU."""
        nobs = np.asarray(nobs, dtype=np.float64)
        L = nobs.size
        nt = np.zeros(L) if n_test is None else np.asarray(n_test, dtype=np.float64)
        cost = nobs ** 3 / 3.0 + nobs ** 2 * nt
        steps = np.ceil(nobs / 128.0)
        group = np.arange(L)
        for j in range(L):
            if op[j] != 0 and src[j] >= 0:
                group[j] = src[j]
        for j in range(L):          # resolve one level of chaining
            group[j] = group[group[j]]
        gcost = np.zeros(L)
        for j in range(L):
            gcost[group[j]] += cost[j] if op[j] == 0 else 2.0 * nobs[j] ** 2
        roots = [g for g in range(L) if group[g] == g]
        roots.sort(key=lambda g: (-gcost[g], g))
        gsteps = np.zeros(L)
        for j in range(L):
            gsteps[group[j]] = max(gsteps[group[j]], steps[j])
        load = np.zeros(world)
        maxsteps = np.zeros(world)
        owner_of_root = {}
        for g in roots:
            t = (load + gcost[g]) / Shard.RATE_FLOPS + Shard.STEP_SECONDS * np.maximum(maxsteps, gsteps[g])
            r = int(np.argmin(t))
            owner_of_root[g] = r
            load[r] += gcost[g]
            maxsteps[r] = max(maxsteps[r], gsteps[g])
        owner = np.array([owner_of_root[group[j]] for j in range(L)], dtype=np.int64)
        return Shard(owner, rank, world)

    # ---- exchange -------------------------------------------------------------------------------
    def _all_gather_padded(self, local, maxlen):
        """All-gather of one padded float64 vector per rank -> list of numpy arrays."""
        import time
        import torch
        td = _pg()
        if td is None:
            raise RuntimeError("sharded model needs an initialised torch.distributed process group")
        t0 = time.perf_counter()
        try:
            return self._all_gather_padded_timed(td, torch, local, maxlen)
        finally:
            self.exchange_seconds += time.perf_counter() - t0
            self.exchanges += 1

    def _all_gather_padded_timed(self, td, torch, local, maxlen):
        backend = _backend(td)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        buf = torch.zeros(maxlen, dtype=torch.float64, device=dev)
        buf[: local.size] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64)).to(dev)
        out = [torch.empty_like(buf) for _ in range(self.world)]
        td.all_gather(out, buf, group=GROUP)
        return [o.cpu().numpy() for o in out]

    def allgather_sum(self, local):
        """Sum over ranks of one fixed-shape float64 array per rank (the partial sums of the predict aggregation:
        W x n_t doubles), as ONE all-gather followed by additions in rank order on every rank -- the same bits
        everywhere, which an all-reduce does not promise."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        if self.world == 1:
            return local.copy()
        parts = self._all_gather_padded(local.reshape(-1), local.size)
        tot = parts[0].copy()
        for p in parts[1:]:
            tot += p
        return tot.reshape(local.shape)

    def gather_leaf_values(self, local_vals):
        """local_vals[i] belongs to leaf self.local[i]; returns the value of every leaf."""
        L = self.owner.size
        local_vals = np.asarray(local_vals, dtype=np.float64)
        if self.world == 1:
            return local_vals.copy()
        counts = np.bincount(self.owner, minlength=self.world)
        parts = self._all_gather_padded(local_vals, int(counts.max()))
        out = np.empty(L)
        for r in range(self.world):
            idx = np.flatnonzero(self.owner == r)
            out[idx] = parts[r][: idx.size]
        return out

    def gather_leaf_columns(self, local_cols):
        """Several per-leaf quantities in ONE collective: local_cols is (n_local_leaves, k); returns (L, k)."""
        local_cols = np.ascontiguousarray(local_cols, dtype=np.float64)
        k = local_cols.shape[1]
        if self.world == 1:
            return local_cols.copy()
        counts = np.bincount(self.owner, minlength=self.world)
        parts = self._all_gather_padded(local_cols.reshape(-1), int(counts.max()) * k)
        out = np.empty((self.owner.size, k))
        for r in range(self.world):
            idx = np.flatnonzero(self.owner == r)
            out[idx] = parts[r][: idx.size * k].reshape(idx.size, k)
        return out

    def gather_ragged_pair(self, a_flat, b_flat, counts):
        """gather_ragged of two vectors with the same ragged layout (mu and var) in ONE collective."""
        if self.world == 1:
            return np.asarray(a_flat, dtype=np.float64).copy(), np.asarray(b_flat, dtype=np.float64).copy()
        both = self.gather_ragged(np.stack([a_flat, b_flat], axis=1).reshape(-1), 2 * np.asarray(counts, dtype=np.int64))
        both = both.reshape(-1, 2)
        return np.ascontiguousarray(both[:, 0]), np.ascontiguousarray(both[:, 1])

    def gather_ragged(self, local_flat, counts):
        """local_flat = concatenation over this rank's leaves (in leaf order) of counts[leaf] values;
        returns the concatenation over ALL leaves in leaf order."""
        counts = np.asarray(counts, dtype=np.int64)
        local_flat = np.asarray(local_flat, dtype=np.float64)
        if self.world == 1:
            return local_flat.copy()
        per_rank = np.array([counts[self.owner == r].sum() for r in range(self.world)])
        parts = self._all_gather_padded(local_flat, int(per_rank.max()))
        ptr = np.concatenate([[0], np.cumsum(counts)])
        out = np.empty(int(ptr[-1]))
        for r in range(self.world):
            pos = 0
            for g in np.flatnonzero(self.owner == r):
                c = int(counts[g])
                out[ptr[g]:ptr[g] + c] = parts[r][pos:pos + c]
                pos += c
        return out
