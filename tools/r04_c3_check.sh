export TMPDIR=/tmp
O=gpurun_out/r04o
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size_configs.py tests/test_determinism.py -m gpu -q --timeout=300 -x 2>&1 | tail -2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/c3_repeat.py > $GRAFT_REPO_ROOT/$O/c3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $O/prof 14 | tee $O/c3_kstats.txt
timeout 300 python tools/c3_repeat.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print([(round(s['wall_sec']*1e3,2), round(s['stats_head'][5]*1e3,3), round(s['t_factor']*1e3,2)) for s in d['solves']])"
