#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}"
export TMPDIR=/tmp
O=gpurun_out/pmc_icache; mkdir -p $O
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python3 tools/wave_check.py --which localization --batch 8192 --reps 1 --skip-generic --out pmc_wave_runs.jsonl > $O/p$i.log 2>&1 < /dev/null
  tail -2 $O/p$i.log | cut -c1-200
done
python3 tools/pmc_summary.py $O/icache.json $O/p1 $O/p2 $O/p3 $O/p4 --kernel ${KERNEL_FILTER:-wave_} > /dev/null
rm -rf $O/p?
cat $O/icache.json
