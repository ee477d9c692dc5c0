"""Where the wall time of a repeated C2 solve through the front-end goes (n = 1e5, Problem.solve(nlp=True,
algorithm="lbfgs") on one Problem object: cached handle): cProfile of 20 calls.  python tools/c2_wall_profile.py"""
import cProfile
import os
import pstats
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dnlp_amd as cp  # noqa: E402
from problem_zoo import rosenbrock_chain  # noqa: E402

warnings.simplefilter("ignore")
p = rosenbrock_chain(cp, 100000)
for _ in range(3):
    p.variables()[0].value = None                       # the default start every time
    p.solve(nlp=True, algorithm="lbfgs")
t = time.time()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    p.variables()[0].value = None
    p.solve(nlp=True, algorithm="lbfgs")
pr.disable()
print("wall per solve ms", 1e3 * (time.time() - t) / 20, "status", p.status, "iterations", p.solver_stats.num_iters)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
