#!/bin/bash
# The committed evidence of a round (GPU box; run from the repo root, outputs under gpurun_out/final/): for EVERY workload
# of the bench line (cfg2, cfg1, cfg4, wide = cfg5) the same bench command under rocprofv3 --kernel-trace --stats with its
# step timeline and the per-launch times without a profiler; for cfg2 the separate --pmc passes (tools/pmc_collect.sh);
# the derived roofline block per workload (tools/roofline_from_profile.py -> profiles/<round>_<workload>_roofline.json,
# which bench.py reads for roofline.in_graph / traffic); the GEMM family against the vendor library; the evaluation's
# kernel table.  Copy gpurun_out/final/* into profiles/ as <round>_*.   ROUND=r04 COMMIT=<hash> bash tools/profile_collect.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final
R=${ROUND:-r06}
rm -rf $O && mkdir -p $O
export DRVAE_SIDE_CUS=64      # fixed split: no tuning replays in the profile
for wl in cfg2 cfg1 cfg4 wide; do
  if [ $wl = wide ]; then ST="--steps 4 --warmup 2"; else ST="--steps 200 --warmup 20"; fi
  # (--no-roofline: the roofline leg re-issues every GEMM launch; the summary must hold the running step's launches only)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$wl -o p -- python3 bench.py --workload $wl $ST --no-cpu-baseline --no-roofline --no-extras --no-steady > $O/${wl}_prof.log 2>&1
  cp $(find $O/k_$wl -name '*kernel_stats.csv' | head -1) $O/${wl}_kernel_stats.csv
  python3 tools/timeline.py $(find $O/k_$wl -name '*kernel_trace.csv' | head -1) > $O/${wl}_step_timeline.txt 2>&1
  rm -rf $O/k_$wl
done
unset DRVAE_SIDE_CUS
# counters for the latency-bound headline AND for the one MFMA-/HBM-bound configuration (cfg 5), incl. the memory-side request pass
export PMC_EXTRA="TCC_EA0_RDREQ_sum,TCC_EA0_RDREQ_LEVEL_sum,TCC_EA0_RDREQ_DRAM_sum,TCC_EA0_RDREQ_32B_sum"
for wl in cfg2 wide; do
  bash tools/pmc_collect.sh $wl > $O/pmc_$wl.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_rr $wl > $O/${wl}_pmc_summary.txt 2>&1
  rm -rf gpurun_out/pmc_rr
done
unset PMC_EXTRA
for wl in cfg2 cfg1 cfg4 wide; do
  if [ $wl = wide ]; then ST="--steps 6 --warmup 2"; else ST="--steps 300 --warmup 20"; fi
  python3 bench.py --workload $wl $ST --no-extras --no-cpu-baseline > $O/${wl}_bench.json 2> $O/${wl}_bench.err
  # every figure of bench.py's roofline block that a profiler has to supply, derived from the files above
  python3 tools/roofline_from_profile.py $O/$wl --commit "${COMMIT:-unknown}" --out $O/${wl}_roofline.json > /dev/null 2>&1
  cp $O/${wl}_roofline.json profiles/${R}_${wl}_roofline.json
  python3 tools/step_profile.py $wl > $O/${wl}_step_isolated.txt 2>&1
done
# the headline once more WITH the roofline files in place: the line to commit as profiles/<round>_cfg2_bench.json
python3 bench.py --steps 300 --warmup 20 > $O/cfg2_bench_final.json 2> $O/cfg2_bench_final.err
( python3 tools/gemm_bench.py --tilings 0; python3 tools/pair_bench.py; python3 tools/heads_bench.py; python3 tools/pipe_check.py --tilings 40 --big --vendor; python3 tools/blas_ref.py ) > $O/gemm_vs_vendor.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_eval -o p -- python3 tools/eval_bench.py 10 > $O/eval_bench.txt 2>&1
python3 tools/kstats.py $O/k_eval 24 > $O/eval_kernel_stats.txt 2>&1
rm -rf $O/k_eval
ls -la $O
