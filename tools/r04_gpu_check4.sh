export TMPDIR=/tmp
O=gpurun_out/r04d
mkdir -p $O
for B in 8192 65536; do
echo "== B=$B"; DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py $B 0 2>&1 | grep "plan:\|kernel_sec\|per iteration" | head -4
done
for W in circle_packing10 power_flow path_planning; do
echo "== $W 1024"; DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py 1024 0 $W 2>&1 | grep "plan:\|kernel_sec\|per iteration\|histogram\|statuses\|Error\|error" | head -6
done
timeout 600 python -m pytest tests/test_determinism.py tests/test_batch_kernel.py tests/test_full_size_configs.py tests/test_sparse_kkt.py tests/test_paper_examples.py tests/test_fused.py -m gpu -q --timeout=300 2>&1 | tail -5
timeout 100 python3 tools/run_paper_examples.py > $O/paper.log 2>&1
python3 - <<PY
import json
for r in json.load(open("gpurun_out/paper_examples.json")):
    print("%-28s st %d it %3d nf %3d lower %.4f solve %.4f factor %.4f obj %.9e" % (r["example"], r["status"], r["iters"], r["factorizations"], r["lower_sec"], r["solve_sec_without_timers"], r["factor_sec"], r["objective"]))
PY
echo "== sparse recovery / phase retrieval / portfolio with kkt_paired_min_n=256"
timeout 100 python3 tools/run_paper_examples.py nb_sparse_recovery nb_phase_retrieval nb_portfolio_construction kkt_paired_min_n=256 > $O/paper2.log 2>&1
python3 - <<PY
import json
for r in json.load(open("gpurun_out/paper_examples.json")):
    print("%-28s st %d it %3d nf %3d lower %.4f solve %.4f factor %.4f obj %.9e" % (r["example"], r["status"], r["iters"], r["factorizations"], r["lower_sec"], r["solve_sec_without_timers"], r["factor_sec"], r["objective"]))
PY
timeout 200 python3 tools/c3_repeat.py 2>&1 | tail -c 700
