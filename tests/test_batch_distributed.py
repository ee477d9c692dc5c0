"""CPU tests of the problem-parallel (N > 1) path: sharding arithmetic and a world_size-2
gloo run whose gathered result must equal the serial one."""
import os
import subprocess
import sys
import warnings

import numpy as np

from batch_problems import build_localization, oracle_solver
from dnlp_amd.batch import gather_rows, shard_bounds, solve_shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything_once():
    for n_items in (0, 1, 7, 8, 1024, 8191):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_bounds(n_items, r, world)
                assert 0 <= lo <= hi <= n_items
                seen += list(range(lo, hi))
            assert seen == list(range(n_items))


def test_serial_batch_recovers_true_positions():
    warnings.simplefilter("ignore")
    rows = solve_shard(build_localization, [0, 1, 2], solver=oracle_solver, workers=1)
    rows = gather_rows(rows, 3)
    assert rows.shape[0] == 3 and np.all(rows[:, 2] == 0)
    assert np.all(np.abs(rows[:, 1]) < 1e-9)    # noise-free: objective 0 at the true position


def test_world_size_2_gloo_matches_serial(tmp_path):
    n_items = 5
    out = str(tmp_path / "rows.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", WORLD_SIZE="2",
               PYTHONWARNINGS="ignore")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), out,
                                       str(n_items)], env=e))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    warnings.simplefilter("ignore")
    ref = solve_shard(build_localization, list(range(n_items)), solver=oracle_solver, workers=1)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got[:, 0], np.arange(n_items))
    np.testing.assert_allclose(got[:, 1:], ref[:, 1:], rtol=1e-12, atol=1e-12, equal_nan=True)
