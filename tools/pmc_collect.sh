#!/bin/bash
# PMC passes over one workload of the bench (separate passes, counters only: no trace domains), per
# MI355X_MICROARCH.md "rocprofv3 PMC slots".   bash tools/pmc_collect.sh [cfg2|cfg1|cfg4|wide]
# Output: gpurun_out/pmc_rr/<pass>/...csv.gz + gpurun_out/pmc_rr/kernel_stats.csv (durations: a --kernel-trace --stats
# run of the SAME command, for the GB/s column of tools/pmc_summary.py).  The program itself follows `--` (no wrappers).
export TMPDIR=/tmp
WL=${1:-cfg2}
# counter collection serializes kernel dispatches: use the single-graph schedule (same kernels; a parked
# device-side wait of the dual-graph schedule could only time out there)
export DRVAE_TUNE=sched=3 DRVAE_SIDE_CUS=0
if [ "$WL" = wide ]; then ST="--steps 3 --warmup 1"; else ST="--steps 20 --warmup 5"; fi
CMD="python3 bench.py --workload $WL $ST --no-cpu-baseline --no-roofline --no-steady --no-extras"
O=gpurun_out/pmc_rr
rm -rf $O && mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $CMD > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/sq -o p -- $CMD > $O/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -o p -- $CMD > $O/tcc.log 2>&1
# memory-side split (what reaches the fabric from L2 vs what the DRAM controllers see): whatever this ROCm lists for gfx950
if [ -n "$PMC_EXTRA" ]; then
  i=0
  for grp in $(echo "$PMC_EXTRA" | tr ';' ' '); do
    rocprofv3 --pmc $(echo $grp | tr ',' ' ') --output-format csv -d $O/extra$i -o p -- $CMD > $O/extra$i.log 2>&1
    i=$((i+1))
  done
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- $CMD > $O/kt.log 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/kt
for f in $O/*/p_counter_collection.csv; do echo $f; head -2 $f | cut -c1-400; done
# keep the upload small
for d in $O/*/; do gzip -f $d/p_counter_collection.csv 2>/dev/null; done
du -sh $O
