export TMPDIR=/tmp
O=gpurun_out/r04m
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/c3_repeat.py > $GRAFT_REPO_ROOT/$O/c3.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $O/c3.log | cut -c1-400
python tools/kstats.py $O/prof 30 | tee $O/c3_kstats.txt
