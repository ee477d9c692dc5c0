# Localization batch: 4 instances per CU either with ~27 KB of vectors + the plan's index arrays in LDS (latency per
# iteration) or with 30 KB of vectors and the plan arrays in global memory (throughput), by batch size.
#   bash tools/micro/batch_lds_share_sweep.sh
for B in 8192 16384 32768 65536; do
  for kb in 27 30; do
    export DNLP_BATCH_VLDS_KB=$kb
    DNLP_BATCH_DEBUG=1 timeout 200 python tools/run_c5_batch.py --batch $B --which localization --check 0 --reps 3 > /tmp/o.txt 2> /tmp/e.txt
    plan=$(grep "per CU" /tmp/e.txt | tail -1 | sed 's/.*dynamic/dynamic/')
    python3 -c "
import json
for l in open('/tmp/o.txt'):
    if l.startswith('{'): d=json.loads(l)
print('batch $B VLDS_KB=$kb', round(d['problems_per_sec_kernel']), 'problems/s (kernel);', '$plan')
"
  done
done
