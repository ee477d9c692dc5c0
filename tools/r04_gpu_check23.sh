export TMPDIR=/tmp
for v in "" "DNLP_BATCH_PACKED=0" "DNLP_BATCH_IDX_LDS=0"; do
echo "== $v"
env $v DNLP_BATCH_DEBUG=1 timeout 200 python tools/batch_tail.py 1024 0 circle_packing10 2>&1 | grep -E "plan:|kernel_sec|per iteration|statuses" | cut -c1-260
done
