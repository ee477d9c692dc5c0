# instruction mix and address-path counters of the batch kernel (separate --pmc passes):  bash tools/r04_pmc_mix.sh
export TMPDIR=/tmp
O=gpurun_out/r04pmc
mkdir -p $O
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_SMEM" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
         "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python3 tools/batch_tail.py 8192 0 > $O/p$i.log 2>&1 < /dev/null
  tail -2 $O/p$i.log | cut -c1-200
done
python3 tools/pmc_summary.py $O/mix.json $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 $O/p7 --kernel batch_solve > /dev/null
rm -rf $O/p?
cat $O/mix.json
