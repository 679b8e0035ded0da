#!/bin/bash
# PMC passes over the cfg-2 bench (separate passes, counters only: no trace domains), per
# MI355X_MICROARCH.md "rocprofv3 PMC slots".  Output: gpurun_out/pmc_rr/<pass>/...csv
export TMPDIR=/tmp
# counter collection serializes kernel dispatches: use the single-graph schedule (same kernels; a parked
# device-side wait of the dual-graph schedule could only time out there)
export DRVAE_TUNE=sched=3 DRVAE_SIDE_CUS=0
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-steady --no-extras"
mkdir -p gpurun_out/pmc_rr
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_rr/fetch -o p -- $CMD > gpurun_out/pmc_rr/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_rr/write -o p -- $CMD > gpurun_out/pmc_rr/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_rr/sq -o p -- $CMD > gpurun_out/pmc_rr/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_rr/tcc -o p -- $CMD > gpurun_out/pmc_rr/tcc.log 2>&1
ls -R gpurun_out/pmc_rr | head -30
for f in gpurun_out/pmc_rr/*/p_counter_collection.csv; do echo $f; head -2 $f | cut -c1-400; done
# keep the upload small
for d in fetch write sq tcc; do gzip -f gpurun_out/pmc_rr/$d/p_counter_collection.csv 2>/dev/null; done
du -sh gpurun_out/pmc_rr
