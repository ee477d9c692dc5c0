"""Problem-parallel batches: independent (parametrised) NLPs sharded over the GPUs of a node.

The nlp=True path shards only across problems: the reference's `best_of`
multistart loop (problems/problem.py:1256-1269) and BASELINE config C5 are independent solves.
Rank r owns the contiguous block [r*ceil(B/W), (r+1)*ceil(B/W)) of instance ids; there is no
data-path collective; ONE exchange at the end gathers {objective, status, iterations, x*}
(torch.distributed all_gather: RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""
from __future__ import annotations

import math
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Sequence, Tuple

import numpy as np


# ---- batched device solve: shared tape structure, per-instance data (csrc/batch.h) -------------
# Order of the per-instance data vector handed to dnlp_solve_batch.
BATCH_DATA_KEYS = ("c0", "c", "b", "Jc", "G_val", "Mg_val", "Mw_val", "MJ_val", "MH_val",
                   "seg_param", "seg_param2", "x0", "lb", "ub", "cl", "cu")


def lower_arrays(problem):
    """Front-end only (no device): Problem -> (tape arrays, data dict, inverse data, flipped)."""
    from .dnlp2smooth import Dnlp2Smooth
    from .nlp_solver import build_nlp_data
    from .problem import Maximize, Minimize, Problem
    flip = isinstance(problem.objective, Maximize)
    if flip:
        problem_min = Problem(Minimize(-problem.objective.expr), problem.constraints)
    else:
        problem_min = problem
    smooth, _ = Dnlp2Smooth().apply(problem_min)
    data, inv = build_nlp_data(smooth, user_variables=problem.variables())
    return data["tape_arrays"], data, inv, flip


def instance_data(arrays) -> np.ndarray:
    return np.concatenate([np.asarray(arrays[k], dtype=np.float64).ravel() for k in BATCH_DATA_KEYS])


def same_structure(a0, a) -> bool:
    """Everything that is not per-instance data (index arrays, segment table, constants of
    quad_forms) must be identical for instances to share one tape."""
    if set(a0) != set(a):
        return False
    for k in a0:
        if k in BATCH_DATA_KEYS:
            if a0[k].shape != a[k].shape:
                return False
        elif not np.array_equal(a0[k], a[k]):
            return False
    return True


def arrays_with_data(arrays0, vec):
    """Inverse of instance_data on the structure of `arrays0` (tests: rebuild an instance's tape)."""
    out = dict(arrays0)
    o = 0
    for k in BATCH_DATA_KEYS:
        n = arrays0[k].size
        out[k] = np.asarray(vec[o:o + n], dtype=np.float64).reshape(arrays0[k].shape)
        o += n
    return out


class BatchResult:
    """Arrays over the batch plus per-variable access in the user's variable shapes."""

    def __init__(self, raw, inv, flip):
        self.x = raw["x"]
        self.obj_val = -raw["obj_val"] if flip else raw["obj_val"]
        self.status = raw["status"]
        self.iterations = raw["iterations"]
        self.factorizations = raw.get("factorizations")
        self.kernel_seconds = raw.get("kernel_seconds")
        self.raw = raw
        self._inv = inv

    def value_of(self, var) -> np.ndarray:
        off = self._inv.var_offsets[var.id]
        flat = self.x[:, off:off + var.size]
        return np.stack([r.reshape(var.shape, order="F") for r in flat]) if var.shape else flat[:, 0]


def _device_handle(arrays, tape, device, opts):
    from . import _capi
    from .nlp_solver import HIPNLP
    from .tape import serialize
    h = _capi.DeviceProblem(serialize(arrays), tape, device=device)
    options = dict(HIPNLP.DEFAULT_OPTIONS)
    # batches: an instance whose static sparse pivots turn singular keeps the sparse factorisation
    # (delta_c) in its first run and may switch to in-kernel Bunch-Kaufman only in the retry rungs —
    # the dense path is ~6x slower per iteration and a few such instances were the tail of a launch
    options["lazy_dense_fallback"] = "yes"
    options.update(opts)
    options.pop("algorithm", None)
    for k, v in options.items():
        h.set_option(k, v)
    return h


def solve_batch(problems: Sequence, device=None, want_duals=False, **opts) -> BatchResult:
    """Solve independent Problems that lower to the same tape structure in one kernel launch.
    Every problem is lowered on the host (the generic path; ParametricBatch avoids that)."""
    lowered = [lower_arrays(p) for p in problems]
    a0, data0, inv0, flip0 = lowered[0]
    for a, _, _, flip in lowered[1:]:
        if flip != flip0 or not same_structure(a0, a):
            raise ValueError("solve_batch: the problems do not share one tape structure")
    mat = np.stack([instance_data(a) for a, _, _, _ in lowered])
    h = _device_handle(a0, data0["tape"], device, opts)
    try:
        raw = h.solve_batch(mat, want_duals=want_duals)
    finally:
        h.close()
    return BatchResult(raw, inv0, flip0)


class ParametricBatch:
    """One Problem written with `Parameter` leaves, many parameter values.

    The tape data of a parametrised DNLP problem is (in all the paper's examples) an affine
    function of the parameter vector: the map is recovered by P + 1 host lowerings (one per
    parameter entry) and CHECKED on a random probe; instances are then generated as one sparse
    matrix product instead of B Python lowerings.  A problem whose data is not affine in its
    parameters (e.g. a product of two parameters) falls back to per-instance lowering."""

    def __init__(self, problem, parameters: Sequence, seed: int = 0):
        import scipy.sparse as sp
        self.problem = problem
        self.params = list(parameters)
        self.sizes = [int(p.size) for p in self.params]
        self.P = int(sum(self.sizes))
        for p in self.params:
            if p.value is None:
                raise ValueError("ParametricBatch: every parameter needs a value (the base point)")
        self.theta0 = self._get()
        self.arrays0, self.data0, self.inv, self.flip = lower_arrays(problem)
        self.d0 = instance_data(self.arrays0)
        cols, rows, vals = [], [], []
        self.affine = True
        for k in range(self.P):
            th = self.theta0.copy()
            th[k] += 1.0
            dk = self._lower_at(th)
            if dk is None:
                self.affine = False
                break
            fin = np.isfinite(dk) & np.isfinite(self.d0)
            if not np.array_equal(dk[~fin], self.d0[~fin]):     # an infinite bound that moves
                self.affine = False
                break
            diff = np.where(fin, dk, 0.0) - np.where(fin, self.d0, 0.0)
            nz = np.flatnonzero(diff)
            rows.extend(nz.tolist())
            cols.extend([k] * nz.size)
            vals.extend(diff[nz].tolist())
        if self.affine:
            self.D = sp.csr_matrix((vals, (rows, cols)), shape=(self.d0.size, self.P))
            rng = np.random.default_rng(seed)
            th = self.theta0 + rng.uniform(-1.0, 1.0, self.P)
            probe = self._lower_at(th)
            pred = self.d0 + self.D @ (th - self.theta0)
            scale = 1.0 + np.abs(probe[np.isfinite(probe)]).max() if probe is not None and probe.size else 1.0
            if probe is None or not np.allclose(pred, probe, rtol=1e-10, atol=1e-10 * scale):
                self.affine = False
        self._set(self.theta0)

    def _get(self):
        return np.concatenate([np.asarray(p.value, dtype=float).reshape(-1, order="F") for p in self.params]) \
            if self.params else np.zeros(0)

    def _set(self, theta):
        o = 0
        for p, n in zip(self.params, self.sizes):
            p.value = np.asarray(theta[o:o + n], dtype=float).reshape(p.shape, order="F")
            o += n

    def _lower_at(self, theta):
        self._set(theta)
        a, _, _, flip = lower_arrays(self.problem)
        if flip != self.flip or not same_structure(self.arrays0, a):
            return None
        return instance_data(a)

    def data(self, thetas) -> np.ndarray:
        """(B, P) parameter values (each row: the parameters flattened F-order, concatenated in
        the order given to the constructor) -> (B, stride) instance data."""
        thetas = np.atleast_2d(np.asarray(thetas, dtype=float))
        if thetas.shape[1] != self.P:
            raise ValueError("expected %d parameter values per instance" % self.P)
        if self.affine:
            delta = thetas - self.theta0[None, :]
            if self.D.shape[0] * self.D.shape[1] <= 8_000_000:
                # small map: one dense GEMM, result already row-major (what the C entry point takes)
                if getattr(self, "_DT_dense", None) is None:
                    self._DT_dense = np.ascontiguousarray(self.D.toarray().T)
                out = delta @ self._DT_dense
                out += self.d0[None, :]
                return out
            return np.ascontiguousarray(self.d0[None, :] + (self.D @ delta.T).T)
        rows = []
        for th in thetas:
            d = self._lower_at(th)
            if d is None:
                raise ValueError("ParametricBatch: an instance changes the tape structure")
            rows.append(d)
        self._set(self.theta0)
        return np.stack(rows)

    def solve(self, thetas, device=None, want_duals=False, warm_from=None, **opts) -> BatchResult:
        """One kernel launch for all rows of `thetas`.  The device handle (tape structure resident
        in HBM) is created on first use and kept for later calls with the same device/options.

        `warm_from`: a BatchResult of the same batch size solved with want_duals=True — every
        instance then starts from that result's primal point and multipliers (IPOPT's
        warm_start_init_point; pass mu_init small, e.g. 1e-6, as with IPOPT)."""
        import time as _t
        t0 = _t.time()
        if warm_from is not None:
            opts.setdefault("warm_start_init_point", "yes")
        self._ensure_handle(device, opts)
        # affine templates: the map lives on the device and a call moves only the parameter rows (the
        # warm-started form patches the rows' x0 block on the host, so it keeps the host-generated rows)
        if self.affine and warm_from is None and hasattr(self._handle, "set_batch_affine_map"):
            thetas = np.atleast_2d(np.asarray(thetas, dtype=float))
            if thetas.shape[1] != self.P:
                raise ValueError("expected %d parameter values per instance" % self.P)
            if not self._map_on_device:
                self._handle.set_batch_affine_map(self.d0, self.theta0, self.D)
                self._map_on_device = True
            t1 = _t.time()
            raw = self._handle.solve_batch(want_duals=want_duals, thetas=thetas)
            if os.environ.get("DNLP_BATCH_DEBUG"):
                print("[batch.py] handle/map %.4f solve_batch(thetas) %.4f" % (t1 - t0, _t.time() - t1), flush=True)
            return BatchResult(raw, self.inv, self.flip)
        mat = self.data(thetas)
        t1 = _t.time()
        warm = None
        if warm_from is not None:
            if "mult_g" not in warm_from.raw:
                raise ValueError("warm_from needs a result solved with want_duals=True")
            o = len(self.d0) - sum(self.arrays0[k].size for k in ("x0", "lb", "ub", "cl", "cu"))
            mat[:, o:o + warm_from.x.shape[1]] = warm_from.x          # the rows' x0 block
            warm = (warm_from.raw["mult_g"], warm_from.raw["mult_x_L"], warm_from.raw["mult_x_U"])
        t2 = _t.time()
        raw = self._handle.solve_batch(mat, want_duals=want_duals, warm=warm)
        t3 = _t.time()
        if os.environ.get("DNLP_BATCH_DEBUG"):
            print("[batch.py] data %.4f handle %.4f solve_batch %.4f" % (t1 - t0, t2 - t1, t3 - t2), flush=True)
        return BatchResult(raw, self.inv, self.flip)

    @staticmethod
    def _solver_options(opts):
        """Solver options of a call, as `solve` sees them: its own named arguments (want_duals, warm_from) taken out, the
        warm-start default applied — the key a handle is cached under must not depend on which entry point was used."""
        o = dict(opts)
        want_duals = bool(o.pop("want_duals", False))
        warm_from = o.pop("warm_from", None)
        if warm_from is not None:
            o.setdefault("warm_start_init_point", "yes")
        return o, want_duals, warm_from

    def _ensure_handle(self, device, opts):
        """(Re)create this object's device handle for (device, options)."""
        key = (device, tuple(sorted((k, str(v)) for k, v in opts.items())))
        if getattr(self, "_handle_key", None) != key:
            self._close_own()
            self._handle = _device_handle(self.arrays0, self.data0["tape"], device, opts)
            self._handle_key = key
            self._map_on_device = False

    def solve_many(self, batches, device=None, in_flight=2, want_duals=False, **opts):
        """A stream of batches with `in_flight` launches overlapping.  A launch lasts as long as its slowest instance —
        evenly spread, the work of an 8192-instance localization batch is ~70 % of the launch — and the workgroups of the
        next launch take the compute units that the tail of the previous one leaves idle.  The overlap lives INSIDE the
        library (include/dnlp_hip.h dnlp_batch_stream_*: every slot its own HIP stream, device buffers and host thread;
        ONE handle, no Python threads — round 4 did this with a handle and a Python thread per worker).  Results are the
        ones `solve` returns, batch by batch, in order, bit for bit."""
        batches = [np.atleast_2d(np.asarray(t, dtype=float)) for t in batches]
        slots = max(1, min(int(in_flight), len(batches)))
        if not self.affine or slots == 1 or opts.get("warm_from") is not None:
            return [self.solve(t, device=device, want_duals=want_duals, **opts) for t in batches]
        opts, _, _ = self._solver_options(opts)
        self._ensure_handle(device, opts)
        if not hasattr(self._handle, "solve_batch_stream") or not hasattr(self._handle.api, "batch_stream_create"):
            return [self.solve(t, device=device, want_duals=want_duals, **opts) for t in batches]
        if not self._map_on_device:
            self._handle.set_batch_affine_map(self.d0, self.theta0, self.D)
            self._map_on_device = True
        raws = self._handle.solve_batch_stream(batches, slots=slots, want_duals=want_duals)
        return [BatchResult(r, self.inv, self.flip) for r in raws]

    def solve_sharded(self, thetas, device=None, force_collective=False, **opts):
        """Problem-parallel solve across the ranks of an initialised torch.distributed group (one
        process per GPU): rank r solves the contiguous block shard_bounds(B, r, W) of the
        rows of `thetas` in ONE launch on its own GPU, then ONE all_gather (RCCL over xGMI with the nccl
        backend) gives every rank all rows {id, objective, status, iterations, x*}.  Without a process
        group it is a plain solve; with a one-rank group the exchange is skipped unless
        `force_collective` asks for it (the single-GPU test of the RCCL path).  Returns (rows ordered by
        instance id, info) with info = ranks the collective saw, bytes it moved, this rank's kernel
        seconds."""
        thetas = np.atleast_2d(np.asarray(thetas, dtype=float))
        B = thetas.shape[0]
        rank, world, backend = 0, 1, None
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank, world, backend = dist.get_rank(), dist.get_world_size(), dist.get_backend()
        except ImportError:       # pragma: no cover
            pass
        lo, hi = shard_bounds(B, rank, world)
        exchanging = backend is not None and (world > 1 or force_collective)
        # with RCCL the shard's rows go into the collective straight from the device buffer the launch packed them in
        # (dnlp_batch_result_rows); the flip of a maximisation's objective is applied on the device tensor
        from_device = exchanging and backend == "nccl" and hi > lo
        if from_device:
            # (the handle key is computed from the solver options alone, as solve() computes it: want_duals / warm_from in
            #  `opts` must not make this a different handle from the one the launch below uses)
            self._ensure_handle(device, self._solver_options(opts)[0])
            from_device = self._handle.keep_batch_result_rows(True)
        local_dev = None
        if hi > lo:
            try:
                res = self.solve(thetas[lo:hi], device=device, **opts)
                local = np.empty((hi - lo, 4 + res.x.shape[1]))           # (filled in place: no temporaries of the 65 536 x 56 rows)
                local[:, 0] = np.arange(lo, hi)
                local[:, 1] = res.obj_val
                local[:, 2] = res.status
                local[:, 3] = res.iterations
                local[:, 4:] = res.x
                ksec = res.kernel_seconds
                launch = res.raw.get("launch")
                if from_device:
                    local_dev = _device_rows(self._handle, lo, -1.0 if self.flip else 1.0)
            finally:
                if from_device:
                    self._handle.keep_batch_result_rows(False)
        else:
            local, ksec, launch = np.zeros((0, 4 + int(self.arrays0["dims"][0]))), 0.0, None
        rows = gather_rows(local, B, force=force_collective, local_dev=local_dev)
        per = math.ceil(B / world) if world > 0 else B
        exchanged = exchanging
        info = {"ranks": world, "backend": backend, "rank": rank, "shard": (lo, hi), "kernel_seconds": ksec,
                "collective": exchanged, "rows_from_device": local_dev is not None,
                "gathered_bytes": int(world * per * rows.shape[1] * 8) if exchanged else 0,
                "launch": launch}        # (this rank's launch form: dnlp_batch_launch_info)
        return rows, info

    def _close_own(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            h.close()
        self._handle, self._handle_key = None, None

    def close(self):
        self._close_own()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    per = math.ceil(n_items / world) if world > 0 else n_items
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def default_solver(problem, **opts):
    """Solve one dnlp_amd Problem on this rank's MI355X; returns (obj, status, iters, x)."""
    chain = problem._build_chain(None)
    data, inv = chain.apply(problem)
    info = chain.solver.solve_via_data(data, True, False, dict(opts))
    data["handle"].close()
    obj = -info["obj_val"] if chain.flip else info["obj_val"]
    return obj, info["status"], info["iterations"], info["x"]


def solve_shard(build: Callable[[int], object], ids: Sequence[int], solver: Callable = default_solver,
                workers: int = 8, **opts) -> np.ndarray:
    """Solve the instances `ids` (build(i) -> Problem).  Returns rows
    [id, objective, status, iterations, x_0 .. x_{N-1}] (N padded to the widest instance).
    Independent solves run on `workers` host threads, each with its own HIP stream."""
    def one(i):
        obj, status, iters, x = solver(build(i), **opts)
        return i, obj, status, iters, np.asarray(x, dtype=np.float64)

    if workers > 1 and len(ids) > 1:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            rows = list(ex.map(one, ids))
    else:
        rows = [one(i) for i in ids]
    width = max([r[4].size for r in rows], default=0)
    out = np.full((len(rows), 4 + width), np.nan)
    for k, (i, obj, status, iters, x) in enumerate(rows):
        out[k, :4] = (i, obj, status, iters)
        out[k, 4:4 + x.size] = x
    return out


class _DeviceRows:
    """A library-owned device buffer of doubles as an object torch can wrap without a copy."""

    def __init__(self, ptr, rows, width):
        self.__cuda_array_interface__ = {"shape": (rows, width), "typestr": "<f8", "data": (ptr, False), "version": 2}


def _device_rows(handle, id0, obj_sign):
    """The last launch's rows {index, objective, status, iterations, x*} as a device tensor (dnlp_batch_result_rows):
    a copy on the device with the shard's first instance id added to column 0 — the library's buffer lives until the
    handle's next launch only."""
    import torch
    ptr, rows, width = handle.batch_result_rows()
    if rows == 0:
        return None
    t = torch.as_tensor(_DeviceRows(ptr, rows, width), device=torch.device("cuda", handle.device)).clone()
    t[:, 0] += float(id0)
    if obj_sign != 1.0:
        t[:, 1] *= obj_sign
    return t


def gather_rows(local: np.ndarray, n_items: int, force: bool = False, local_dev=None):
    """The single exchange of the path: every rank receives all rows, ordered by instance id.
    Works with any initialised torch.distributed backend; a no-op without a process group.  A
    one-rank group skips the collective unless `force` is set (then the all_reduce + all_gather run
    on the backend all the same: how the RCCL path is exercised on a single MI355X).
    `local_dev`: this rank's rows as a device tensor (the launch's own buffer, _device_rows) — then nothing of this
    rank's contribution passes through host memory on its way into the collective."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:   # pragma: no cover
        return local
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return local[np.argsort(local[:, 0])] if local.size else local
    world = dist.get_world_size()
    per = math.ceil(n_items / world)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    # rows are padded to a common width and a common count so one all_gather suffices
    width = torch.tensor([local.shape[1] if local.size else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(width, op=dist.ReduceOp.MAX)
    w = int(width.item())
    buf = torch.full((per, w), float("nan"), dtype=torch.float64, device=dev)
    if local_dev is not None and dev.type == "cuda":
        buf[:local_dev.shape[0], :local_dev.shape[1]] = local_dev
    elif local.size:
        buf[:local.shape[0], :local.shape[1]] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    allrows = torch.cat(parts).cpu().numpy()
    allrows = allrows[~np.isnan(allrows[:, 0])]
    return allrows[np.argsort(allrows[:, 0])]
