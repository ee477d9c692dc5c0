#!/bin/bash
# kernel timeline of one blocked LDL^T at a given order: tools/ldlt_trace.sh N1 M [rows of the timeline]
ROOT=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$ROOT"
N1=${1:-10000}; M=${2:-1000}; R=${3:-60}
O=$ROOT/gpurun_out/prof_ld
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $ROOT/tools/time_ldlt.py $N1 $M 3 > /dev/null 2>&1 )
python3 tools/kstats.py $O 10
T=$(ls $O/*/*_kernel_trace.csv | tail -1); python3 tools/ldlt_timeline.py $T $R | cut -c1-400; rm -rf $O
