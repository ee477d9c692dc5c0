export TMPDIR=/tmp
O=gpurun_out/r04i
mkdir -p $O
for B in 8192 65536; do
echo "== B=$B"; DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py $B 0 2>&1 | grep "plan:\|kernel_sec\|per iteration" | head -4
done
echo "== circle packing (n=4) 8192"; DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py 8192 0 circle_packing 2>&1 | grep "plan:\|kernel_sec\|per iteration" | head -4
for W in circle_packing10 power_flow path_planning; do
echo "== $W 1024"; timeout 120 python3 tools/batch_tail.py 1024 0 $W 2>&1 | grep "kernel_sec\|per iteration" | head -4
done
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -x 2>&1 | tail -5
timeout 100 python3 tools/run_paper_examples.py > $O/paper.log 2>&1
python3 - <<PY
import json
for r in json.load(open("gpurun_out/paper_examples.json")):
    print("%-28s st %d it %3d nf %3d lower %.4f solve %.4f factor %.4f obj %.9e" % (r["example"], r["status"], r["iters"], r["factorizations"], r["lower_sec"], r["solve_sec_without_timers"], r["factor_sec"], r["objective"]))
PY
timeout 200 python3 tools/c3_repeat.py 2>&1 | tail -c 300
