export TMPDIR=/tmp
O=gpurun_out/r04g
mkdir -p $O
echo "== C2 n sweep"
timeout 300 python3 tools/c2_device_loop.py 100000 200000 300000 600000 1000000 2>&1 | cut -c1-330 | tee $O/c2_sweep.jsonl
timeout 300 python -m pytest tests/test_fused.py -m gpu -q --timeout=300 2>&1 | tail -3
echo "== bench, short, cpu baseline"
timeout 900 python3 bench.py --steps 1 --warmup 0 --no-full-solve > $O/bench_short.json 2> $O/bench_short.err
python3 - <<PY
import json
d=json.loads(open("$O/bench_short.json").read().strip().splitlines()[-1])
cb=d["cpu_baseline"]
print(d["value"], d["roofline"]["frac"])
for r in cb.get("sweep",[]):
    print(r["n"], {k:(round(v["s_per_factorization"],4), round(v["factorization_gflops"],1)) for k,v in r.items() if isinstance(v,dict) and "s_per_factorization" in v}, r.get("blocked_phase_seconds_last_factorization"))
print({k:cb.get(k) for k in ("value","cores","factorization","host_cores","blocked_s_per_factorization_by_threads","blocked_threads","sample")})
PY
tail -3 $O/bench_short.err
