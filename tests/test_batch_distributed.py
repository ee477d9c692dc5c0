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


def test_world_size_2_gloo_parametric_batch_shards(tmp_path):
    """The product's own sharding + gather (ParametricBatch.solve_sharded) across two ranks: every
    instance of the gathered result must be the instance's own solution (noise-free localization:
    objective 0 at the true position, test_nlp_solvers.py:175-189), ordered by instance id."""
    n_items = 7                       # odd: the last rank's shard is shorter than the others'
    out = str(tmp_path / "rows_pb.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29573", WORLD_SIZE="2", PYTHONWARNINGS="ignore")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), out,
                                       str(n_items), "parametric"], env=e))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    assert got.shape[0] == n_items
    np.testing.assert_array_equal(got[:, 0], np.arange(n_items))
    assert np.all(got[:, 2] == 0) and np.all(np.abs(got[:, 1]) < 1e-9)
    import batch_problems as bp
    prob, params, sample, xvar = bp.template_localization()
    from dnlp_amd.batch import ParametricBatch
    pb = ParametricBatch(prob, params)
    off = pb.inv.var_offsets[xvar.id]
    for i in range(n_items):
        x_true = np.random.default_rng(i).uniform(-3, 3, 2)
        assert np.allclose(got[i, 4 + off:4 + off + 2], x_true, atol=1e-5)
