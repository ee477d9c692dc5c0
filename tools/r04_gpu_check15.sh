export TMPDIR=/tmp
O=gpurun_out/r04l
mkdir -p $O
for i in 1 2 3; do timeout 600 python -m pytest "tests/test_determinism.py" -m gpu -q --timeout=300 -k "nmf or phase" 2>&1 | tail -1; done
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -x > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
timeout 300 python tools/c3_repeat.py 2>&1 | tail -6
