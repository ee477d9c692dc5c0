#!/bin/bash
# Counter passes of the C5 bench command (separate --pmc passes, kernel trace only), summarised into the profile file that
# bench.py's roofline.traffic reads:   tools/pmc_c5.sh [round tag, default r05] [template] [batch]
#   -> gpurun_out/<tag>_pmc_wave_<batch>.json   (copy to profiles/)
cd "${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}"
export TMPDIR=/tmp
TAG=${1:-r06}
W=${2:-localization}
B=${3:-8192}
MINMS=${4:-13}          # launches at least this long are the B-instance launches (the command also times shards of 1024: 7-9 ms)
O=gpurun_out/${TAG}_pmc_c5
mkdir -p $O
CMD="python3 bench.py --workload c5 --which $W --batch $B --steps 3 --warmup 1 --no-cpu"
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/c5_$tag -- $CMD > $O/c5_$tag.log 2>&1 < /dev/null
done
python3 tools/pmc_summary.py $O/raw.json $O/c5_SQ_WAVE_CYCLES $O/c5_SQ_INSTS_VALU $O/c5_FETCH_SIZE $O/c5_WRITE_SIZE --kernel wave_ --min-ms $MINMS > /dev/null
python3 - "$O/raw.json" "gpurun_out/${TAG}_pmc_wave_${B}.json" "$W" "$B" "$CMD" "$MINMS" <<'PY'
import json, sys
raw = json.load(open(sys.argv[1]))
name = max(raw, key=lambda k: raw[k].get("SQ_WAVE_CYCLES_per_dispatch", 0))
r = raw[name]
out = {"which": sys.argv[3], "batch": int(sys.argv[4]),
       "command": "rocprofv3 --pmc {SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY | SQ_INSTS_* | FETCH_SIZE | WRITE_SIZE} --kernel-trace -- "
                  + sys.argv[5] + "  (tools/pmc_c5.sh; launches of at least " + sys.argv[6] + " ms — the 8192-instance ones: warm-up, fresh batches, re-solve passes, two-in-flight passes)",
       "kernel": name,
       "fetch_bytes_x2_corrected_per_launch": r.get("FETCH_bytes_x2_corrected_per_dispatch"),
       "write_bytes_per_launch": r.get("WRITE_bytes_per_dispatch"),
       "traffic_bytes_per_launch": (r.get("FETCH_bytes_x2_corrected_per_dispatch") or 0) + (r.get("WRITE_bytes_per_dispatch") or 0),
       "wave_cycles_waiting": r.get("SQ_WAIT_ANY_frac_of_wave_cycles"), "wave_cycles_issuing": r.get("SQ_ACTIVE_INST_ANY_frac_of_wave_cycles"),
       "wave_cycles_issue_stalled": r.get("SQ_WAIT_INST_ANY_frac_of_wave_cycles"),
       "instructions_per_launch": {k[len("SQ_INSTS_"):-len("_per_dispatch")]: r[k] for k in r if k.startswith("SQ_INSTS_") and k.endswith("_per_dispatch")},
       "avg_launch_ms_under_counters": r.get("avg_ms_under_SQ_WAVE_CYCLES"), "raw": {name: r}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "raw"}, indent=1))
PY
rm -rf $O/c5_*
