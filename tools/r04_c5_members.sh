# the other three C5 members at 1024 fresh instances per step:  bash tools/r04_c5_members.sh
export TMPDIR=/tmp
mkdir -p gpurun_out/r04m5
for W in circle_packing10 path_planning power_flow; do
  timeout 300 python3 bench.py --workload c5 --which $W --batch 1024 --steps 4 --warmup 1 --no-cpu 2>/dev/null | grep "^{" > gpurun_out/r04m5/bench_c5_${W}_1024.json
  python3 -c "
import json
d=json.loads(open('gpurun_out/r04m5/bench_c5_${W}_1024.json').read()); c=d['config']; print('$W', round(d['value'],1), round(d['ms_per_step'],1), c['optimal'], round(c['two_batches_in_flight_problems_per_s'] or 0,1), round(c['resolve_same_batch_problems_per_s'],1))"
done
