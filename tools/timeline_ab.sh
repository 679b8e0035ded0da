cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/tl; rm -rf $O; mkdir -p $O
export DRVAE_SIDE_CUS=64
for v in "$@"; do
export $v
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o p -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > $O/prof_$v.log 2>&1
python3 tools/timeline.py $(find $O/k -name '*kernel_trace.csv' | head -1) > $O/timeline_$v.txt 2>&1
rm -rf $O/k
done
