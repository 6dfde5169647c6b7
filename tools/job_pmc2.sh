# cache-path counters of the fused message kernel (development): one rocprofv3 --pmc pass per small group
export TMPDIR=/tmp
out=gpurun_out/pmc2; mkdir -p $out
i=0
for grp in \
 "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
 "TCP_TCP_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
 "TD_TC_STALL_sum TD_TD_BUSY_sum" \
 "TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" ; do
  i=$((i+1))
  echo "pass $i: $grp" >> $out/progress.txt
  N=500000 ITERS=2 timeout -k 5 75 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- python3 tools/msg_micro.py > $out/pass$i.log 2>&1 || echo "pass $i failed" >> $out/progress.txt
done
python3 tools/pmc_summary.py $out msg_fused > $out/summary.txt 2>&1
