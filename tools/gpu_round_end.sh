# End-of-round measurements on the GPU box (one gpurun call):  bash tools/gpu_round_end.sh
export TMPDIR=/tmp
O=gpurun_out/r03end
mkdir -p $O
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/prof_bench.log 2>&1 < /dev/null
python3 tools/kstats.py $O/prof_bench > $O/bench_kernel_stats.txt 2>/dev/null
head -12 $O/bench_kernel_stats.txt
rm -rf $O/prof_bench
timeout 300 python3 bench.py --workload c5 --batch 8192 --steps 5 --warmup 2 > $O/bench_c5_8192.json 2>/dev/null
tail -c 1500 $O/bench_c5_8192.json
timeout 200 python3 tools/c3_repeat.py > $O/c3_repeat.log 2>&1; tail -c 600 $O/c3_repeat.log
timeout 200 python3 tools/first_call_breakdown.py > $O/first_call_breakdown.jsonl 2>/dev/null; cut -c1-330 $O/first_call_breakdown.jsonl
