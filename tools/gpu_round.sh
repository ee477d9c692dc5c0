#!/bin/bash
# Measurements of a round on the GPU box, one part (or several) per call; everything lands in gpurun_out/<TAG>/ and what is
# to be judged is copied into profiles/ as <TAG>_<name> (profiles/README.md lists the part behind every file).
#   tools/gpu_round.sh TAG part [part ...]
# parts
#   suite      python -m pytest tests -m gpu -q                               -> gpu_suite_tail.txt
#   bench      python bench.py (the default C4 line) + rocprofv3 kernel table -> bench_n100000.json, bench_n100000_kernel_stats.txt
#   pmc_c4     FETCH_SIZE / WRITE_SIZE / MFMA-busy passes of the bench command -> pmc_bench_traffic.json
#   c5         bench.py --workload c5 at 8192 (with the CPU baseline) and 65536 + kernel table
#   members    the other C5 members at 1024 fresh instances per step          -> c5_members.jsonl
#   pmc_c5     tools/pmc_c5.sh TAG (counter passes of the C5 bench command)    -> <TAG>_pmc_wave_8192.json
#   wave       wavefront solver against the generic kernel, every template; phase profile (tools/wave_profile.sh)
#   stream     batches in flight, 1 .. 4 (tools/c5_in_flight.py)               -> c5_in_flight.json
#   c3         C3 through one handle, kernel table, timeline, LDL^T by order
#   misc       first-call breakdown, C2 end to end + by n, the nine paper examples
#   pmc_mix    instruction mix / LDS / address-path counters of the batch kernel (tools/pmc_wave.sh)
#   ab_c2_barrier, ab_c4_sweep   the two A/B measurements of round 4 (C2 grid barrier three ways; C4 sweeps vs step kernels)
#   ab_ldlt_fused                the round-5 A/B of the 512-column panel forms of the blocked LDL^T (tools/ldlt_fused_ab.sh)
#   pmc_icache, launch_breakdown instruction-fetch / LDS-wait counters of the wavefront kernel; where a batch launch's wall time goes
#   spec, spec_prof, wg, pmc_wg  round 6: the per-template kernels against the library's own (tools/wave_spec_check.py, wave_wg_check.py), their phase profiles and counters
ROOT=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$ROOT"
export TMPDIR=/tmp
TAG=${1:?tag}; shift
O=gpurun_out/$TAG
mkdir -p $O
for PART in "$@"; do
case $PART in
suite)
  timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log | tee $O/gpu_suite_tail.txt ;;
bench)
  ( time timeout 900 python3 bench.py ) > $O/bench_n100000.json 2> $O/bench_default.err; tail -c 2500 $O/bench_n100000.json; tail -4 $O/bench_default.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-full-solve > $O/prof_bench.log 2>&1 < /dev/null
  python3 tools/kstats.py $O/prof_bench > $O/bench_n100000_kernel_stats.txt 2>/dev/null; head -8 $O/bench_n100000_kernel_stats.txt; rm -rf $O/prof_bench ;;
pmc_c4)
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    t=$(echo $C | cut -d' ' -f1)
    timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/bench_$t -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-full-solve > $O/bench_$t.log 2>&1 < /dev/null
  done
  python3 tools/pmc_summary.py $O/pmc_bench_traffic_raw.json $O/bench_FETCH_SIZE $O/bench_WRITE_SIZE $O/bench_SQ_VALU_MFMA_BUSY_CYCLES --kernel gemm_nt_update_fast --update-queue > /dev/null
  python3 tools/pmc_c4_post.py $O/pmc_bench_traffic_raw.json $O/pmc_bench_traffic.json 100000 > /dev/null      # (the schema bench.py reads)
  rm -rf $O/bench_FETCH_SIZE $O/bench_WRITE_SIZE $O/bench_SQ_VALU_MFMA_BUSY_CYCLES; head -c 1500 $O/pmc_bench_traffic.json ;;
c5)
  timeout 600 python3 bench.py --workload c5 --batch 8192 --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/c5_bench_8192.json; tail -c 2500 $O/c5_bench_8192.json
  timeout 300 python3 bench.py --workload c5 --batch 65536 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 > $O/c5_bench_65536.json; tail -c 1200 $O/c5_bench_65536.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 bench.py --workload c5 --batch 8192 --steps 3 --warmup 1 --no-cpu > $O/prof_c5.log 2>&1 < /dev/null
  python3 tools/kstats.py $O/prof_c5 > $O/c5_kernel_stats.txt 2>/dev/null; head -6 $O/c5_kernel_stats.txt; rm -rf $O/prof_c5 ;;
members)
  : > $O/c5_members.jsonl
  for W in circle_packing circle_packing10 path_planning power_flow; do
    timeout 600 python3 bench.py --workload c5 --which $W --batch 1024 --steps 4 --warmup 1 2>/dev/null | grep "^{" | tail -1 >> $O/c5_members.jsonl
  done
  # (circle packing n = 10 again at 8192 instances: from 1536 on it runs the workgroup kernel's second form — two wavefronts per
  #  instance, four workgroups per compute unit)
  timeout 600 python3 bench.py --workload c5 --which circle_packing10 --batch 8192 --steps 3 --warmup 1 --no-cpu 2>/dev/null | grep "^{" | tail -1 >> $O/c5_members.jsonl
  python3 -c "
import json
for l in open('$O/c5_members.jsonl'):
    d=json.loads(l); c=d['config']; print(d['metric'][28:80], round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'optimal', c['optimal'], 'in flight', round(c['two_batches_in_flight_problems_per_s'] or 0,1), (c.get('kernel_form') or {}).get('wave_form'))" ;;
pmc_c5)
  tools/pmc_c5.sh $TAG localization 8192 | tail -25 ;;
wave)
  rm -f gpurun_out/wave_check.jsonl gpurun_out/prof.jsonl
  tools/wave_profile.sh localization 8192 2>&1 | tee $O/wave_phase_profile_localization_8192.txt | grep -c profile
  timeout 900 python tools/wave_check.py --which localization,circle_packing,circle_packing10 --batch 1024 --reps 3 > /dev/null 2>&1
  DNLP_BATCH_WAVE=2 timeout 900 python tools/wave_check.py --which path_planning,power_flow --batch 1024 --reps 2 > /dev/null 2>&1
  cp gpurun_out/wave_check.jsonl $O/wave_check.jsonl; cut -c1-400 $O/wave_check.jsonl ;;
stream)
  timeout 600 python tools/c5_in_flight.py 8192 2>/dev/null | tail -1 > $O/c5_in_flight.json
  timeout 600 python tools/c5_in_flight.py 2048 2>/dev/null | tail -1 >> $O/c5_in_flight.json; cat $O/c5_in_flight.json ;;
c3)
  timeout 200 python3 tools/c3_repeat.py > $O/c3_repeat.log 2>&1; tail -c 900 $O/c3_repeat.log; cp gpurun_out/c3_repeat.json $O/c3_repeat.json
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof_c3 -- python3 $ROOT/tools/c3_repeat.py > $ROOT/$O/prof_c3.log 2>&1 )
  python3 tools/kstats.py $O/prof_c3 30 > $O/c3_kernel_stats.txt; head -12 $O/c3_kernel_stats.txt
  T=$(ls $O/prof_c3/*/*_kernel_trace.csv | tail -1)
  python3 tools/kernel_order.py $T 19 > $O/c3_kernel_order.txt 2>/dev/null; python3 tools/ldlt_timeline.py $T 400 > $O/c3_timeline.txt 2>/dev/null; rm -rf $O/prof_c3
  for sz in "1500 500" "3000 1000" "5000 1000" "10000 1000" "14000 2000" "20000 2000"; do timeout 300 python tools/time_ldlt.py $sz 5 2>&1 | tail -1 | cut -c1-120; done > $O/ldlt_by_order.jsonl; cat $O/ldlt_by_order.jsonl ;;
misc)
  timeout 300 python3 tools/first_call_breakdown.py > $O/first_call_breakdown.jsonl 2>/dev/null; cut -c1-330 $O/first_call_breakdown.jsonl
  timeout 300 python3 tools/run_c2_end_to_end.py 100000 2>/dev/null | grep "^{" > $O/c2_end_to_end_n100000.json; tail -c 700 $O/c2_end_to_end_n100000.json
  timeout 600 python tools/c2_device_loop.py 100000 200000 300000 600000 1000000 2>/dev/null | grep "^{" > $O/c2_n_sweep.jsonl; cut -c1-330 $O/c2_n_sweep.jsonl
  timeout 600 python3 tools/run_paper_examples.py > $O/paper_examples_gpu.json 2>/dev/null; tail -c 1500 $O/paper_examples_gpu.json ;;
pmc_mix)
  tools/pmc_wave.sh localization 8192 $TAG | tail -40 ;;
ab_c2_barrier)
  for v in "" "DNLP_LBFGS_ATOMIC_SUMS=1" "DNLP_LBFGS_ATOMIC_SUMS=1 DNLP_LBFGS_FULL_FENCE=1"; do
    echo "== $v"; env $v timeout 300 python tools/c2_device_loop.py 100000 200000 300000 600000 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['n'], round(d['best']['device_loop_ms'],3), round(d['us_per_slot'],2), d['best']['status'], d['best']['iterations'])"
  done | tee $O/c2_barrier_ab.txt ;;
ab_c4_sweep)
  for v in "DNLP_LDLT_SWEEP_MAX_BLOCKS=256" ""; do
    echo "== ${v:-sweeps up to 1024 blocks}"; env $v timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-full-solve 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  done | tee $O/c4_sweep_ab.txt ;;
ab_ldlt_fused)
  bash tools/ldlt_fused_ab.sh 2>&1 | tee $O/ldlt_fused_ab.txt ;;
pmc_icache)
  bash tools/pmc_icache.sh > /dev/null 2>&1; cp gpurun_out/pmc_icache/icache.json $O/pmc_wave_icache.json; head -c 600 $O/pmc_wave_icache.json ;;
launch_breakdown)
  for b in 1024 8192; do python tools/c5_launch_breakdown.py localization $b 2>&1 | tail -7; done | tee $O/c5_launch_breakdown.txt ;;
spec)
  # per-template kernel (csrc/wave_codegen.h) against the library's own wavefront kernel: bits / path, kernel time, us per iteration
  rm -f gpurun_out/wave_spec_check.jsonl
  timeout 600 python tools/wave_spec_check.py --which localization,circle_packing,circle_packing10 --batch 1024,8192 --reps 3 > /dev/null 2>&1
  timeout 300 python tools/wave_spec_check.py --which localization --batch 65536 --reps 2 > /dev/null 2>&1
  cp gpurun_out/wave_spec_check.jsonl $O/wave_spec_check.jsonl
  python3 -c "
import json
for l in open('$O/wave_spec_check.jsonl'):
    d = json.loads(l); print(d['problem'], d['batch'], 'spec ms %.2f own ms %.2f | us/it %.1f vs %.1f | k/s %.1f vs %.1f | same iterations %d of %d' % (d['spec_kernel_ms_best'], d['own_kernel_ms_best'], d['spec_instance_us_per_iter'], d['own_instance_us_per_iter'], d['spec_problems_per_s_kernel']/1e3, d['own_problems_per_s_kernel']/1e3, d['same_iterations'], d['batch']), d['spec_launch']['wave_form'], d['spec_launch']['wave_spec'])" ;;
spec_prof)
  # cycle profile of the per-template kernels' phases (DNLP_WAVE_SPEC_PROF: the kernel is compiled with the counters of wave_ipm.h)
  DNLP_WAVE_SPEC=1 DNLP_WAVE_SPEC_PROF=1 timeout 300 python tools/wave_check.py --which localization --batch 8192 --reps 2 --skip-generic --out prof_spec.jsonl 2>&1 | grep "wave profile" > $O/wave_phase_profile_spec_localization_8192.txt
  for W in path_planning power_flow; do
    DNLP_WAVE_SPEC=1 DNLP_WAVE_SPEC_PROF=1 timeout 300 python tools/wave_wg_check.py --which $W --batch 256 --reps 1 --skip-generic --out prof_wg.jsonl 2>&1 | grep "wave profile" > $O/wave_phase_profile_wg_${W}_256.txt
  done
  head -30 $O/wave_phase_profile_spec_localization_8192.txt ;;
wg)
  # workgroup-per-instance kernel (csrc/wave_wg_kernel.h) against the generic batch kernel: path planning, power flow
  rm -f gpurun_out/wave_wg_check.jsonl
  timeout 900 python tools/wave_wg_check.py --which path_planning,power_flow --batch 1024 --reps 2 > /dev/null 2>&1
  cp gpurun_out/wave_wg_check.jsonl $O/wave_wg_check.jsonl
  python3 -c "
import json
for l in open('$O/wave_wg_check.jsonl'):
    d = json.loads(l); print(d['problem'], 'wg k/s %.2f (%.3f ms/it) generic k/s %.2f (%.3f ms/it) same status %d same iterations %d' % (d['wg_problems_per_s_kernel']/1e3, d['wg_instance_ms_per_iter'], d['generic_problems_per_s_kernel']/1e3, d['generic_instance_ms_per_iter'], d['same_status'], d['same_iterations']), d['wg_status_hist'], d['generic_status_hist'])"
  timeout 300 python tools/pf_status_hist.py > $O/power_flow_status_hist.txt 2>/dev/null; cat $O/power_flow_status_hist.txt ;;
pmc_wg)
  for W in path_planning power_flow; do bash tools/pmc_wg.sh $W 1024 $TAG/pmc_wg_$W 2>&1 | tail -2; cp gpurun_out/$TAG/pmc_wg_$W/wg_counters.json $O/pmc_wg_$W.json; done ;;
*) echo "unknown part $PART" ;;
esac
done
