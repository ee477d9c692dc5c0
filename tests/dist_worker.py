"""Worker for the world_size-2 gloo tests of the problem-parallel path (launched by
test_batch_distributed.py with RANK / WORLD_SIZE / MASTER_* set).

mode "shard"      : solve_shard + gather_rows with the CPU oracle as the per-problem solver
mode "parametric" : ParametricBatch.solve_sharded — the product's sharding + gather code — with the
                    per-rank launch answered by the CPU oracle (no GPU in this container)
mode "device"     : ParametricBatch.solve_sharded on the MI355X of this rank with the backend named by
                    DNLP_TEST_BACKEND (nccl = RCCL); with WORLD_SIZE=1 the exchange is forced through the
                    backend all the same (force_collective) so that RCCL init + all_reduce + all_gather on
                    device tensors run on a single GPU"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch.distributed as dist  # noqa: E402

import batch_problems as bp  # noqa: E402
from batch_problems import build_localization, oracle_solver  # noqa: E402
from dnlp_amd import batch as batch_mod  # noqa: E402
from dnlp_amd.batch import gather_rows, shard_bounds, solve_shard  # noqa: E402


class OracleBatchHandle:
    """Stands in for the device handle of a rank: answers solve_batch row by row with the CPU oracle."""

    def __init__(self, arrays, tape, device, opts):
        self.arrays, self.opts = arrays, dict(opts)
        self.n, self.m = int(arrays["dims"][0]), int(arrays["dims"][1])

    def solve_batch(self, mat, want_duals=False, warm=None):
        from dnlp_amd.batch import arrays_with_data
        from dnlp_amd.nlp_solver import HIPNLP
        from dnlp_amd.tape import serialize
        from oracle.oracle_capi import OracleProblem
        B = mat.shape[0]
        out = {"x": np.zeros((B, self.n)), "obj_val": np.zeros(B), "status": np.zeros(B, np.int32),
               "iterations": np.zeros(B, np.int32), "factorizations": np.zeros(B, np.int32), "kernel_seconds": 0.0,
               "phase_seconds": np.zeros((B, 4))}
        for i in range(B):
            a = arrays_with_data(self.arrays, mat[i])
            o = OracleProblem(serialize(a))
            for k, v in HIPNLP.DEFAULT_OPTIONS.items():
                o.set_option(k, v)
            r = o.solve(a["x0"])
            out["x"][i], out["obj_val"][i], out["status"][i], out["iterations"][i] = r["x"], r["obj_val"], r["status"], r["iterations"]
        return out

    def close(self):
        pass


def main():
    out_path, n_items = sys.argv[1], int(sys.argv[2])
    mode = sys.argv[3] if len(sys.argv) > 3 else "shard"
    backend = os.environ.get("DNLP_TEST_BACKEND", "gloo")
    if backend == "nccl":
        import torch
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode == "device":
        prob, params, sample, _ = bp.template_localization()
        pb = batch_mod.ParametricBatch(prob, params)
        thetas = np.stack([sample(i) for i in range(n_items)])
        allrows, info = pb.solve_sharded(thetas, force_collective=True)
        assert info["ranks"] == world and info["backend"] == backend and info["collective"]
        assert info["gathered_bytes"] == world * -(-n_items // world) * allrows.shape[1] * 8
        if backend == "nccl":
            # the rows went into the collective from the launch's own device buffer (dnlp_batch_result_rows) and are the
            # rows a plain solve of the same instances returns through host memory
            assert info["rows_from_device"]
            res = pb.solve(thetas)
            host = np.concatenate([np.arange(n_items, dtype=float)[:, None], res.obj_val[:, None], res.status[:, None].astype(float),
                                   res.iterations[:, None].astype(float), res.x], axis=1)
            assert np.array_equal(allrows, host)
        # gather_rows by itself, a second time, on rows it did not produce
        again = gather_rows(allrows[shard_bounds(n_items, rank, world)[0]:shard_bounds(n_items, rank, world)[1]],
                            n_items, force=True)
        assert np.array_equal(again, allrows)
        pb.close()
    elif mode == "parametric":
        batch_mod._device_handle = lambda arrays, tape, device, opts: OracleBatchHandle(arrays, tape, device, opts)
        prob, params, sample, _ = bp.template_localization()
        pb = batch_mod.ParametricBatch(prob, params)
        thetas = np.stack([sample(i) for i in range(n_items)])
        allrows, info = pb.solve_sharded(thetas)
        assert info["ranks"] == world and info["backend"] == "gloo" and info["gathered_bytes"] > 0
        assert info["shard"] == shard_bounds(n_items, rank, world)
    else:
        lo, hi = shard_bounds(n_items, rank, world)
        local = solve_shard(build_localization, list(range(lo, hi)), solver=oracle_solver, workers=1)
        allrows = gather_rows(local, n_items)
    if rank == 0:
        np.save(out_path, allrows)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
