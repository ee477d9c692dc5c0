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


def gather_rows(local: np.ndarray, n_items: int):
    """The single exchange of the path: every rank receives all rows, ordered by instance id.
    Works with any initialised torch.distributed backend; a no-op without a process group."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:   # pragma: no cover
        return local
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local[np.argsort(local[:, 0])] if local.size else local
    world = dist.get_world_size()
    per = math.ceil(n_items / world)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) \
        if dist.get_backend() == "nccl" else torch.device("cpu")
    # rows are padded to a common width and a common count so one all_gather suffices
    width = torch.tensor([local.shape[1] if local.size else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(width, op=dist.ReduceOp.MAX)
    w = int(width.item())
    buf = torch.full((per, w), float("nan"), dtype=torch.float64, device=dev)
    if local.size:
        buf[:local.shape[0], :local.shape[1]] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    allrows = torch.cat(parts).cpu().numpy()
    allrows = allrows[~np.isnan(allrows[:, 0])]
    return allrows[np.argsort(allrows[:, 0])]
