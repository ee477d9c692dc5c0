"""Worker for the world_size-2 gloo test of the problem-parallel path (launched by
test_batch_distributed.py with RANK / WORLD_SIZE / MASTER_* set)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch.distributed as dist  # noqa: E402

from batch_problems import build_localization, oracle_solver  # noqa: E402
from dnlp_amd.batch import gather_rows, shard_bounds, solve_shard  # noqa: E402


def main():
    out_path, n_items = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(n_items, rank, world)
    local = solve_shard(build_localization, list(range(lo, hi)), solver=oracle_solver, workers=1)
    allrows = gather_rows(local, n_items)
    if rank == 0:
        np.save(out_path, allrows)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
