#!/bin/bash
# Instruction mix, wait / issue cycles and HBM traffic of the wavefront batch kernel (separate --pmc passes):
#   tools/pmc_wave.sh [template] [batch] [tag]      -> gpurun_out/<tag>/wave_mix.json
cd "${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}"
export TMPDIR=/tmp
W=${1:-localization}
B=${2:-8192}
O=gpurun_out/${3:-pmc_wave}
mkdir -p $O
i=0
if [ "$4" = "quick" ]; then
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python3 tools/wave_check.py --which $W --batch $B --reps 1 --skip-generic --out pmc_wave_runs.jsonl > $O/p$i.log 2>&1 < /dev/null
  done
  python3 tools/pmc_summary.py $O/wave_quick.json $O/p1 $O/p2 $O/p3 --kernel ${KERNEL_FILTER:-wave_} > /dev/null
  rm -rf $O/p?
  python3 -c "
import json,sys
d=json.load(open('$O/wave_quick.json'))
for k,v in d.items():
    print({a:b for a,b in v.items() if a.endswith('per_dispatch') or 'frac' in a})
"
  exit 0
fi
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_SMEM" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python3 tools/wave_check.py --which $W --batch $B --reps 1 --skip-generic --out pmc_wave_runs.jsonl > $O/p$i.log 2>&1 < /dev/null
  tail -1 $O/p$i.log | cut -c1-160
done
python3 tools/pmc_summary.py $O/wave_mix.json $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 $O/p7 --kernel ${KERNEL_FILTER:-wave_} > /dev/null
rm -rf $O/p?
cat $O/wave_mix.json
