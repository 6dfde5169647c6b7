#!/bin/bash
# usage: tools/pmc_passes.sh <outdir> <program> [args...]   — one rocprofv3 --pmc run per counter group
# (counters only: never combined with sys/hip/hsa tracing — gpurun refuses that mix)
out=$1; shift
export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_WAVES" \
 "FETCH_SIZE GRBM_GUI_ACTIVE" \
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -- "$@" > $out.pass$i.log 2>&1 || echo "pass $i failed"
done
