# round-4 check 2: rest of the GPU suite, batch kernel variants
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout=300 --durations=8 > $O/gpu_suite.log 2>&1
tail -25 $O/gpu_suite.log
for B in 8192 65536; do
echo "== baseline B=$B"; timeout 120 python3 tools/batch_tail.py $B 0 2>&1 | grep -v "^\[" | head -8
echo "== WPE=2 default plan B=$B"; DNLP_BATCH_WPE=2 DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py $B 0 2>&1 | grep "plan:\|kernel_sec\|per iteration" | head -4
for KB in 12 16 20 24; do
echo "== WPE=2 VLDS_KB=$KB B=$B"; DNLP_BATCH_WPE=2 DNLP_BATCH_VLDS_KB=$KB DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py $B 0 2>&1 | grep "plan:\|kernel_sec\|per iteration" | head -4
done
echo "== WPE=1 VLDS_KB=16 B=$B"; DNLP_BATCH_VLDS_KB=16 DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py $B 0 2>&1 | grep "plan:\|kernel_sec\|per iteration" | head -4
done
for W in circle_packing10 power_flow path_planning; do
echo "== $W 1024 baseline"; timeout 120 python3 tools/batch_tail.py 1024 0 $W 2>&1 | grep "kernel_sec\|per iteration\|histogram\|statuses" | head -5
echo "== $W 1024 WPE=2"; DNLP_BATCH_WPE=2 timeout 120 python3 tools/batch_tail.py 1024 0 $W 2>&1 | grep "kernel_sec\|per iteration" | head -3
done
