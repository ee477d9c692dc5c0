export TMPDIR=/tmp
O=gpurun_out/r04j
mkdir -p $O
echo "== c3 repeat (sweep kernels)"; timeout 200 python3 tools/c3_repeat.py 2>&1 | tail -c 620
echo "== c3 repeat (step kernels)"; DNLP_LDLT_SWEEP=0 timeout 200 python3 tools/c3_repeat.py 2>&1 | tail -c 320
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size_configs.py tests/test_determinism.py tests/test_paired_kkt.py tests/test_sparse_kkt.py tests/test_paper_examples.py -m gpu -q --timeout=300 > $O/tests.log 2>&1; tail -4 $O/tests.log; grep -n "Error\|assert " $O/tests.log | head
for n in 3000 6000 22000; do timeout 100 python3 tools/time_ldlt.py $n 3 2>&1 | tail -2; done
