export TMPDIR=/tmp
out=gpurun_out/pmc3; mkdir -p $out
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
 "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
 "TCP_TCP_LATENCY_sum TCP_GATE_EN2_sum" ; do
  i=$((i+1))
  echo "pass $i: $grp" >> $out/progress.txt
  N=500000 ITERS=2 timeout -k 5 90 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- python3 tools/msg_micro.py > $out/pass$i.log 2>&1 || echo "pass $i failed" >> $out/progress.txt
done
python3 tools/pmc_summary.py $out msg_fused > $out/summary.txt 2>&1
