export TMPDIR=/tmp
O=gpurun_out/r04t
mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/time_ldlt.py 10000 1000 3 > $GRAFT_REPO_ROOT/$O/log.txt 2>&1 )
tail -1 $O/log.txt | cut -c1-100
T=$(ls $O/prof/*/*_kernel_trace.csv | tail -1)
python3 tools/kernel_order.py $T 13.5 > $O/order.txt 2>/dev/null; tail -1 $O/order.txt
awk '{ for(i=1;i<=NF;i++) if ($i=="before") { g=$(i+1); if (g+0 > 8) print g, $0 } }' $O/order.txt | sort -rn | head -30 | cut -c1-150
rm -rf $O/prof
