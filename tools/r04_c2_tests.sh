export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused.py tests/test_fused_codegen.py tests/test_determinism.py -m gpu -q --timeout=600 -x -k "c2 or persist or lbfgs or fused or generated" 2>&1 | tail -6
timeout 300 python tools/c2_device_loop.py 100000 300000 2>&1 | tail -2 | cut -c1-400
DNLP_LBFGS_COOPERATIVE=0 timeout 300 python tools/c2_device_loop.py 100000 2>&1 | tail -1 | cut -c1-300
