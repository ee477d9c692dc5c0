# PMC passes of the bench command itself (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one) and of the
# generated C2 kernel at n = 1e8.  Run on the GPU box:  bash tools/pmc_bench.sh
export TMPDIR=/tmp
mkdir -p gpurun_out/r02g
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/r02g/pmc_bench_$tag -- python3 bench.py --steps 1 --warmup 0 --no-cpu > gpurun_out/r02g/pmc_bench_$tag.log 2>&1 < /dev/null
done
python3 tools/pmc_summary.py gpurun_out/r02g/pmc_bench.json gpurun_out/r02g/pmc_bench_FETCH_SIZE gpurun_out/r02g/pmc_bench_WRITE_SIZE gpurun_out/r02g/pmc_bench_SQ_VALU_MFMA_BUSY_CYCLES --kernel gemm_nt_update_fast --update-queue | head -40
# the XCD-aware 8 x 8 super-tile order (DNLP_LDLT_XCD=1, off by default): traffic and speed of the same command
DNLP_LDLT_XCD=1 timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02g/pmc_bench_xcd_FETCH_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu > gpurun_out/r02g/pmc_bench_xcd_FETCH.log 2>&1 < /dev/null
python3 tools/pmc_summary.py gpurun_out/r02g/pmc_bench_xcd.json gpurun_out/r02g/pmc_bench_xcd_FETCH_SIZE --kernel gemm_nt_update_fast --update-queue | head -20
DNLP_LDLT_XCD=1 timeout 300 python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/r02g/bench_xcd.json 2> gpurun_out/r02g/bench_xcd.err < /dev/null
timeout 300 python3 bench.py --steps 2 --warmup 1 --no-cpu > gpurun_out/r02g/bench_default.json 2> gpurun_out/r02g/bench_default.err < /dev/null
python3 -c "
import json
for f in ('bench_default','bench_xcd'):
    d=json.load(open('gpurun_out/r02g/%s.json'%f)); print(f, d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])
"
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/r02g/pmc_c2_$C -- python3 tools/c2_kernel_only.py 100000000 5 > gpurun_out/r02g/pmc_c2_$C.log 2>&1 < /dev/null
done
python3 tools/pmc_summary.py gpurun_out/r02g/pmc_c2.json gpurun_out/r02g/pmc_c2_FETCH_SIZE gpurun_out/r02g/pmc_c2_WRITE_SIZE --kernel dnlp_fused_eval | head -30
