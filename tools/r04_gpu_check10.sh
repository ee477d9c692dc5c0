export TMPDIR=/tmp
O=gpurun_out/r04h
mkdir -p $O
for B in 8192 65536; do
echo "== packed B=$B"; DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py $B 0 2>&1 | grep "plan:\|kernel_sec\|per iteration\|statuses" | head -5
echo "== regular B=$B"; DNLP_BATCH_PACKED=0 DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py $B 0 2>&1 | grep "plan:\|kernel_sec\|per iteration\|statuses" | head -5
done
echo "== circle packing (n=4) 8192"; DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py 8192 0 circle_packing 2>&1 | grep "plan:\|kernel_sec\|per iteration\|statuses" | head -5
DNLP_BATCH_PACKED=0 DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py 8192 0 circle_packing 2>&1 | grep "plan:\|kernel_sec\|per iteration\|statuses" | head -5
for W in circle_packing10 power_flow path_planning; do
echo "== $W 1024"; DNLP_BATCH_DEBUG=1 timeout 120 python3 tools/batch_tail.py 1024 0 $W 2>&1 | grep "plan:\|kernel_sec\|per iteration" | head -4
done
timeout 900 python -m pytest tests/test_batch_kernel.py tests/test_determinism.py tests/test_full_size_configs.py tests/test_warm_start.py tests/test_sparse_kkt.py tests/test_appendix_d.py -m gpu -q --timeout=300 2>&1 | tail -6
