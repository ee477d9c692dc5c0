export TMPDIR=/tmp
O=gpurun_out/r04k
mkdir -p $O
timeout 600 python -m pytest "tests/test_determinism.py" -m gpu -q --timeout=300 -k "nmf or phase" > $O/det.log 2>&1; grep -n "assert\|Error\|differ\|passed\|failed" $O/det.log | head -20
DNLP_LDLT_SWEEP=0 timeout 600 python -m pytest "tests/test_determinism.py" -m gpu -q --timeout=300 -k "nmf or phase" 2>&1 | tail -2
