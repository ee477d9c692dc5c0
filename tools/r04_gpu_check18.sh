export TMPDIR=/tmp
O=gpurun_out/r04n
mkdir -p $O
for P in 1 2 3; do
echo "== split $P"
DNLP_LDLT_SWEEP_SPLIT=$P timeout 600 python -m pytest tests/test_determinism.py tests/test_gpu_parity.py tests/test_full_size_configs.py -m gpu -q --timeout=300 -k "nmf or phase or blocked or c3 or ldlt" 2>&1 | tail -2
DNLP_LDLT_SWEEP_SPLIT=$P timeout 300 python tools/c3_repeat.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print([(round(s['wall_sec']*1e3,2), round(s['stats_head'][5]*1e3,3)) for s in d['solves']])"
done
