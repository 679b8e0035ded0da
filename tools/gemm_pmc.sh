#!/bin/bash
# PMC passes (counters only, no trace domains) over one GEMM shape: gemm_pmc.sh <tag> M N K akc bkc tiling
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/gpmc_$TAG
mkdir -p $OUT
P=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" \
           "MfmaUtil MeanOccupancyPerActiveCU" "LdsUtil MemUnitStalled" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
  P=$((P+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$P -o p -- python3 tools/gemm_one.py "$@" > $OUT/p$P.log 2>&1
done
python3 tools/pmc_avg.py $(find $OUT -name 'p_counter_collection.csv') --match gemm_kernel > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name '*.csv' -delete
