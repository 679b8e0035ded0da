# same-box A/B of two builds of the library over the default bench: bash tools/ab_lib.sh build_lab/libdrvae_prev.so [extra bench args]
# (alternates the two, three rounds: the boxes' run-to-run spread is ~0.5 %)
prev=$1; shift
for i in 1 2 3; do
for lib in "$prev" drvae_amd/libdrvae_hip.so; do
env DRVAE_HIP_LIB=$lib python bench.py --steps 3000 --warmup 50 --no-cpu-baseline --no-roofline --no-extras --no-steady "$@" 2>&1 | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); print('$lib', r['ms_per_step'], r['losses_last_step']['ELBO'])"
done
done
