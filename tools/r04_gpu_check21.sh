export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_batch_kernel.py -m gpu -q --timeout=600 -x -k "order_600" --durations=3 2>&1 | tail -15
