#!/bin/bash
# Wait / issue / instruction-fetch counters of the workgroup-per-instance batch kernel (separate --pmc passes):
#   tools/pmc_wg.sh [template] [batch] [tag]      -> gpurun_out/<tag>/wg_counters.json
cd "${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}"
export TMPDIR=/tmp
export DNLP_WAVE_SPEC=1
W=${1:-path_planning}
B=${2:-1024}
O=gpurun_out/${3:-pmc_wg}
mkdir -p $O
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python3 tools/wave_wg_check.py --which $W --batch $B --reps 1 --skip-generic --out pmc_wg_runs.jsonl > $O/p$i.log 2>&1 < /dev/null
done
python3 tools/pmc_summary.py $O/wg_counters.json $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 --kernel dnlp_wave_wg > /dev/null
rm -rf $O/p?
python3 -c "
import json
d = json.load(open('$O/wg_counters.json'))
for k, v in d.items():
    print(k); print({a: b for a, b in v.items() if a.endswith('per_dispatch') or 'frac' in a})
"
