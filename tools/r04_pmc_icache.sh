# instruction-cache counters of the batch kernel (separate --pmc passes):  bash tools/r04_pmc_icache.sh
export TMPDIR=/tmp
O=gpurun_out/r04pmc
mkdir -p $O
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/q$i -- python3 tools/batch_tail.py 8192 0 > $O/q$i.log 2>&1 < /dev/null
  tail -2 $O/q$i.log | cut -c1-200
done
python3 tools/pmc_summary.py $O/icache.json $O/q1 $O/q2 $O/q3 $O/q4 --kernel batch_solve > /dev/null
rm -rf $O/q?
cat $O/icache.json
