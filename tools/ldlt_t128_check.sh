# A/B of the 128-column sub-panel path of the blocked LDL^T against the 32-column chain (GPU box).
mkdir -p gpurun_out/t128
for T in 1 0; do
  for sz in "10000 1000" "1200 300" "3000 841" "300 100" "20000 2000" "6000 0"; do
    echo "T128=$T $sz: $(DNLP_LDLT_T128=$T timeout 300 python3 tools/time_ldlt.py $sz 3 2>&1 | tail -1)"
  done
done 2>&1 | tee gpurun_out/t128/ab.txt
