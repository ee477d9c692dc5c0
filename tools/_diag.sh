cd $GRAFT_REPO_ROOT
timeout 500 python tools/wave_wg_check.py --which path_planning,power_flow --batch 1024 --reps 2 --out t.jsonl 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,dict)})
    else: print(l.rstrip()[:300])"
for W in path_planning power_flow; do
  DNLP_WAVE_SPEC=1 DNLP_WAVE_SPEC_PROF=1 timeout 300 python tools/wave_wg_check.py --which $W --batch 256 --reps 1 --skip-generic --out prof_wg.jsonl 2>&1 | grep "wave profile" | grep -E "solve \(all\)|ldl_solve|ldl_factor|iterations"
done
