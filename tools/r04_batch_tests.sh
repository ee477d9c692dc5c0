export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_batch_kernel.py tests/test_sparse_kkt.py tests/test_limited_memory.py -m gpu -q --timeout=900 -x --durations=6 2>&1 | grep -v "^Starting\|^Solving" | tail -25
