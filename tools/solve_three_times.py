import sys, os, time, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dnlp_amd as cp
from paper_examples import PAPER, PAPER_LARGE
d = dict(PAPER); d.update(PAPER_LARGE)
name = sys.argv[1]
prob = d[name](cp)
chain = prob._build_chain(None)
data, inv = chain.apply(prob)
for rep in range(3):
    t0 = time.time()
    info = chain.solver.solve_via_data(data, True, False, {})
    print(json.dumps({"example": name, "rep": rep, "solve_sec": time.time() - t0, "iters": int(info["iterations"]), "nf": int(info["stats"][1]), "t_factor": float(info["stats"][4]), "t_solve": float(info["stats"][5]), "t_eval": float(info["stats"][3])}))
