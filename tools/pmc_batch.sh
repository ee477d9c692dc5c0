export TMPDIR=/tmp
mkdir -p gpurun_out/r02e
for B in 8192 65536; do
  for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $C | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/r02e/pmc_b${B}_$tag -- python3 tools/run_c5_batch.py --batch $B --which localization --check 0 --reps 1 > gpurun_out/r02e/pmc_b${B}_$tag.log 2>&1 < /dev/null
  done
  python3 tools/pmc_summary.py gpurun_out/r02e/pmc_batch_$B.json gpurun_out/r02e/pmc_b${B}_SQ_WAVE_CYCLES gpurun_out/r02e/pmc_b${B}_FETCH_SIZE gpurun_out/r02e/pmc_b${B}_WRITE_SIZE --kernel batch_solve > /dev/null
done
timeout 300 python3 tools/run_c5_batch.py --batch 8192 --which localization,circle_packing --check 16 --reps 3 2>&1 | grep "^{" | cut -c1-900
timeout 300 python3 tools/run_c5_batch.py --batch 65536 --which localization --check 0 --reps 2 2>&1 | grep "^{" | cut -c1-700
cat gpurun_out/r02e/pmc_batch_8192.json
