# C4 (n = 1e5) with the one-launch triangular sweeps (782 blocks) against the step kernels:  bash tools/r04_c4_sweep_ab.sh
export TMPDIR=/tmp
for v in "DNLP_LDLT_SWEEP_MAX_BLOCKS=256" ""; do
  echo "== ${v:-sweeps up to 1024 blocks}"
  env $v timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-full-solve 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=500 -x -k "c4_full_size" 2>&1 | tail -2
