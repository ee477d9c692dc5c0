# round-4 check 1: GPU suite (incl. determinism), C5 bench (fresh batches), paper examples timing
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --timeout=300 --durations=8 > $O/gpu_suite.log 2>&1
tail -25 $O/gpu_suite.log
timeout 300 python3 bench.py --workload c5 --batch 8192 --steps 5 --warmup 2 --no-cpu > $O/bench_c5_8192.json 2>$O/bench_c5_8192.err
tail -c 1800 $O/bench_c5_8192.json
timeout 200 python3 tools/run_paper_examples.py > $O/paper.log 2>&1
python3 - <<PY
import json
for r in json.load(open("gpurun_out/paper_examples.json")):
    print("%-28s st %d it %3d nf %3d lower %.4f solve %.4f factor %.4f obj %.9e" % (r["example"], r["status"], r["iters"], r["factorizations"], r["lower_sec"], r["solve_sec_without_timers"], r["factor_sec"], r["objective"]))
PY
