export TMPDIR=/tmp
O=gpurun_out/r04f
mkdir -p $O
echo "== host blocked LDLT phases (GPU box host)"
for T in 64 32; do DNLP_HOST_LDLT_TIMING=1 OMP_NUM_THREADS=$T timeout 300 python3 tools/host_ldlt_timing.py 10000 $T 2>&1 | tail -5; done
echo "== C2 n sweep (persistent kernel modes)"
timeout 300 python3 tools/c2_device_loop.py 100000 300000 1000000 3000000 2>&1 | cut -c1-420
echo "== C2 forced modes at 1e5"
DNLP_LBFGS_PERSIST_MODE=1 timeout 100 python3 tools/c2_device_loop.py 100000 2>&1 | cut -c1-300
DNLP_LBFGS_PERSIST_MODE=2 timeout 100 python3 tools/c2_device_loop.py 100000 2>&1 | cut -c1-300
echo "== first call profile c3"
DNLP_PROFILE_APPLY=1 timeout 200 python3 tools/first_call_breakdown.py c3 2>&1 | tail -30
timeout 600 python -m pytest tests/test_fused.py tests/test_determinism.py -m gpu -q --timeout=300 2>&1 | tail -4
