#!/bin/bash
# A/B of the 512-column panel forms of the blocked LDL^T (csrc/ldlt_blocked.h): sub-panel chain / fused rows / + one-launch diagonal block
for sz in "1500 500" "3000 1000" "5000 1000" "10000 1000" "14000 2000" "17000 2000"; do
  for f in "0 0" "1 0" "1 1"; do
    set -- $f
    echo -n "fused_rows=$1 diag512=$2 $sz: "; DNLP_LDLT_FUSED_ROWS=$1 DNLP_LDLT_DIAG512=$2 timeout 300 python tools/time_ldlt.py $sz 5 2>&1 | tail -1 | cut -c1-150
  done
done
