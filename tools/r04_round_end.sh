# End-of-round-4 measurements on the GPU box (one gpurun call):  bash tools/r04_round_end.sh [part]
export TMPDIR=/tmp
O=gpurun_out/r04end
mkdir -p $O
PART=${1:-all}
if [ $PART = all ] || [ $PART = suite ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
fi
if [ $PART = all ] || [ $PART = bench ]; then
  ( time timeout 900 python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err
  tail -c 2500 $O/bench_default.json; tail -4 $O/bench_default.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-full-solve > $O/prof_bench.log 2>&1 < /dev/null
  python3 tools/kstats.py $O/prof_bench > $O/bench_kernel_stats.txt 2>/dev/null; head -8 $O/bench_kernel_stats.txt
  rm -rf $O/prof_bench
fi
if [ $PART = all ] || [ $PART = c5 ]; then
  timeout 300 python3 bench.py --workload c5 --batch 8192 --steps 5 --warmup 2 > $O/bench_c5_8192.json 2>/dev/null; tail -c 1800 $O/bench_c5_8192.json
  timeout 300 python3 bench.py --workload c5 --batch 65536 --steps 3 --warmup 1 --no-cpu > $O/bench_c5_65536.json 2>/dev/null; tail -c 1200 $O/bench_c5_65536.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 bench.py --workload c5 --batch 8192 --steps 3 --warmup 1 --no-cpu > $O/prof_c5.log 2>&1 < /dev/null
  python3 tools/kstats.py $O/prof_c5 > $O/c5_kernel_stats.txt 2>/dev/null; head -6 $O/c5_kernel_stats.txt
  rm -rf $O/prof_c5
  timeout 600 python3 tools/dense_batch_orders.py 16 2>/dev/null | grep kkt_order > $O/dense_batch_orders.jsonl; cat $O/dense_batch_orders.jsonl
  timeout 300 python3 tools/dense600_best_of.py 64 2>/dev/null | grep -v "^Starting" | cut -c1-200 > $O/dense600_best_of_64.txt; cat $O/dense600_best_of_64.txt
fi
if [ $PART = all ] || [ $PART = c3 ]; then
  timeout 200 python3 tools/c3_repeat.py > $O/c3_repeat.log 2>&1; tail -c 900 $O/c3_repeat.log
  cp gpurun_out/c3_repeat.json $O/c3_repeat.json
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c3 -- python3 $GRAFT_REPO_ROOT/tools/c3_repeat.py > $GRAFT_REPO_ROOT/$O/prof_c3.log 2>&1 )
  python3 tools/kstats.py $O/prof_c3 30 > $O/c3_kernel_stats.txt; head -12 $O/c3_kernel_stats.txt
  T=$(ls $O/prof_c3/*/*_kernel_trace.csv | tail -1)
  python3 tools/kernel_order.py $T 19 > $O/c3_kernel_order.txt 2>/dev/null; tail -1 $O/c3_kernel_order.txt
  python3 tools/ldlt_timeline.py $T 400 > $O/c3_timeline.txt 2>/dev/null; tail -1 $O/c3_timeline.txt | cut -c1-200
  rm -rf $O/prof_c3
  for sz in "1500 500" "3000 1000" "5000 1000" "10000 1000" "14000 2000" "20000 2000"; do timeout 300 python tools/time_ldlt.py $sz 5 2>&1 | tail -1 | cut -c1-120; done > $O/ldlt_by_order.jsonl; cat $O/ldlt_by_order.jsonl
fi
if [ $PART = all ] || [ $PART = misc ]; then
  timeout 300 python3 tools/first_call_breakdown.py > $O/first_call_breakdown.jsonl 2>/dev/null; cut -c1-330 $O/first_call_breakdown.jsonl
  timeout 300 python3 tools/run_c2_end_to_end.py 100000 > $O/c2_end_to_end_n100000.json 2>/dev/null; tail -c 700 $O/c2_end_to_end_n100000.json
  timeout 600 python3 tools/run_paper_examples.py > $O/paper_examples.json 2>/dev/null; tail -c 1500 $O/paper_examples.json
fi
