# Counter passes of the C5 bench command on the final build (separate --pmc passes, kernel trace only):  bash tools/pmc_c5_final.sh
export TMPDIR=/tmp
O=gpurun_out/r03pmc_final
mkdir -p $O
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/c5_$tag -- python3 bench.py --workload c5 --batch 8192 --steps 3 --warmup 1 --no-cpu > $O/c5_$tag.log 2>&1 < /dev/null
done
python3 tools/pmc_summary.py $O/pmc_c5_8192.json $O/c5_SQ_WAVE_CYCLES $O/c5_FETCH_SIZE $O/c5_WRITE_SIZE --kernel batch_solve > /dev/null
rm -rf $O/c5_*
cat $O/pmc_c5_8192.json | head -60
