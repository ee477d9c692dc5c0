export TMPDIR=/tmp
O=gpurun_out/r04bench
mkdir -p $O
( time timeout 900 python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 6000 $O/bench_default.json; tail -5 $O/bench_default.err
timeout 300 python3 bench.py --workload c5 --batch 8192 --no-cpu > $O/bench_c5_8192.json 2>/dev/null; tail -c 1500 $O/bench_c5_8192.json
