#!/bin/bash
# The committed evidence of a round (GPU box; run from the repo root, outputs under gpurun_out/final/):
# the default bench line, the same command under rocprofv3 --kernel-trace --stats (cfg 2 and the wide
# configuration, with their step timelines) and the PMC passes (tools/pmc_collect.sh).  Copy the summaries into
# profiles/ as rNN_*.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final
rm -rf $O && mkdir -p $O
python3 bench.py --steps 300 --warmup 20 > $O/cfg2_bench.json 2> $O/cfg2_bench.err
python3 bench.py --workload wide --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-steady > $O/wide_bench.json 2> $O/wide_bench.err
export DRVAE_SIDE_CUS=64      # fixed split: no tuning replays in the profile
# (--no-roofline: the roofline leg re-issues every GEMM launch; the summary must hold the running step's launches only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k2 -o p -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-extras --no-steady > $O/cfg2_prof.log 2>&1
cp $(find $O/k2 -name '*kernel_stats.csv' | head -1) $O/cfg2_kernel_stats.csv
python3 tools/timeline.py $(find $O/k2 -name '*kernel_trace.csv' | head -1) > $O/cfg2_step_timeline.txt 2>&1
rm -rf $O/k2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kw -o p -- python3 bench.py --workload wide --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-steady > $O/wide_prof.log 2>&1
cp $(find $O/kw -name '*kernel_stats.csv' | head -1) $O/wide_kernel_stats.csv
python3 tools/timeline.py $(find $O/kw -name '*kernel_trace.csv' | head -1) > $O/wide_step_timeline.txt 2>&1
rm -rf $O/kw
unset DRVAE_SIDE_CUS
bash tools/pmc_collect.sh > $O/pmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_rr > $O/cfg2_pmc_summary.txt 2>&1
rm -rf gpurun_out/pmc_rr
# every figure of bench.py's roofline block that a profiler has to supply, derived from the files above
python3 tools/roofline_from_profile.py $O/cfg2 --out $O/cfg2_roofline.json > /dev/null 2>&1
# ... and the bench line once more WITH that file in place (its roofline.frac is read from profiles/<round>_cfg2_roofline.json):
# this is the line to commit as profiles/<round>_cfg2_bench.json
cp $O/cfg2_roofline.json profiles/${ROUND:-r03}_cfg2_roofline.json
python3 bench.py --steps 300 --warmup 20 > $O/cfg2_bench_final.json 2> $O/cfg2_bench_final.err
python3 tools/step_profile.py cfg2 > $O/cfg2_step_isolated.txt 2>&1
python3 tools/step_profile.py wide > $O/wide_step_isolated.txt 2>&1
python3 tools/gemm_bench.py --tilings 0 > $O/gemm_ours.txt 2>&1
python3 tools/blas_ref.py > $O/gemm_vendor.txt 2>&1
ls -la $O
