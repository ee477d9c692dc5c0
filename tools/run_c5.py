"""BASELINE config C5: a batch of parametrised paper NLPs sharded over the GPUs of one node,
one process per GPU, one RCCL all_gather at the end.

    python tools/run_c5.py --per-gpu 64                       # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 ... tools/run_c5.py --per-gpu 1024
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from batch_problems import build_circle_packing, build_localization  # noqa: E402
from dnlp_amd.batch import gather_rows, shard_bounds, solve_shard  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-gpu", type=int, default=64)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--family", default="localization")
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("DNLP_DEVICE", str(local))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    build = {"localization": build_localization, "circle_packing": build_circle_packing}[args.family]
    n_items = args.per_gpu * world
    lo, hi = shard_bounds(n_items, rank, world)
    t0 = time.time()
    rows = solve_shard(build, list(range(lo, hi)), workers=args.workers)
    t_solve = time.time() - t0
    allrows = gather_rows(rows, n_items)
    dt = time.time() - t0
    if rank == 0:
        ok = int(np.sum(allrows[:, 2] == 0))
        out = {"family": args.family, "n_gpus": world, "instances": n_items, "optimal": ok,
               "seconds": dt, "problems_per_sec": n_items / dt, "iters_total": float(np.sum(allrows[:, 3])),
               "iters_per_sec": float(np.sum(allrows[:, 3])) / dt, "solve_seconds_rank0": t_solve,
               "gather_bytes": int(allrows.nbytes), "workers": args.workers}
        print(json.dumps(out))
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c5_%s_%dgpu.json" % (args.family, world)), "w"), indent=1)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
