export TMPDIR=/tmp
O=gpurun_out/r04c2
mkdir -p $O
timeout 600 python tools/c2_device_loop.py 100000 200000 300000 600000 1000000 2>/dev/null | grep "^{" > $O/c2_n_sweep.jsonl; cut -c1-330 $O/c2_n_sweep.jsonl
timeout 300 python tools/run_c2_end_to_end.py 100000 2>/dev/null | grep "^{" > $O/c2_end_to_end_n100000.json; cat $O/c2_end_to_end_n100000.json
timeout 300 python tools/c2_wall_profile.py 2>&1 | head -1
timeout 900 python -m pytest tests/test_fused.py tests/test_fused_codegen.py tests/test_determinism.py tests/test_gpu_parity.py -m gpu -q --timeout=600 -x 2>&1 | tail -2
