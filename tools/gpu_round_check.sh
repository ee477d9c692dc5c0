timeout 200 python tools/run_paper_examples.py > gpurun_out/paper_tail.log 2>&1
python - <<PY
import json
for r in json.load(open("gpurun_out/paper_examples.json")):
    print("%-28s st %d it %3d nf %3d lower %.4f solve %.4f factor %.4f obj %.9e" % (r["example"], r["status"], r["iters"], r["factorizations"], r["lower_sec"], r["solve_sec"], r["factor_sec"], r["objective"]))
PY
timeout 600 python -m pytest tests -m gpu -q --durations=5 --timeout=200 2>&1 | tail -12
