# Round-4 counter passes (separate --pmc passes, kernel trace only): the bench command (C4) and the C5 bench command; then the
# C5 members.  Run on the GPU box:  bash tools/r04_pmc.sh
export TMPDIR=/tmp
O=gpurun_out/r04pmc
mkdir -p $O
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/bench_$tag -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-full-solve > $O/bench_$tag.log 2>&1 < /dev/null
done
python3 tools/pmc_summary.py $O/pmc_bench.json $O/bench_FETCH_SIZE $O/bench_WRITE_SIZE $O/bench_SQ_VALU_MFMA_BUSY_CYCLES --kernel gemm_nt_update_fast --update-queue > /dev/null
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/c5_$tag -- python3 bench.py --workload c5 --batch 8192 --steps 3 --warmup 1 --no-cpu > $O/c5_$tag.log 2>&1 < /dev/null
done
python3 tools/pmc_summary.py $O/pmc_c5_8192.json $O/c5_SQ_WAVE_CYCLES $O/c5_FETCH_SIZE $O/c5_WRITE_SIZE --kernel batch_solve > /dev/null
rm -rf $O/bench_* $O/c5_*
for W in circle_packing10 power_flow path_planning; do
  B=1024
  timeout 300 python3 bench.py --workload c5 --which $W --batch $B --steps 3 --warmup 1 --no-cpu 2>/dev/null | grep "^{" > $O/bench_c5_${W}_$B.json
done
ls -la $O; head -c 1500 $O/pmc_bench.json; echo; head -c 1500 $O/pmc_c5_8192.json
