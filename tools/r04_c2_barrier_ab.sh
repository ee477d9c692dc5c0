export TMPDIR=/tmp
# C2 grid barrier three ways on one box (profiles/r04_c2_barrier_ab.txt):  bash tools/r04_c2_barrier_ab.sh
for v in "" "DNLP_LBFGS_ATOMIC_SUMS=1" "DNLP_LBFGS_ATOMIC_SUMS=1 DNLP_LBFGS_FULL_FENCE=1"; do
echo "== $v"
env $v timeout 300 python tools/c2_device_loop.py 100000 200000 300000 600000 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n'], round(d['best']['device_loop_ms'],3), round(d['us_per_slot'],2), d['best']['status'], d['best']['iterations'])"
done
