# A/B of one environment variable over the default bench: bash tools/ab_env.sh NAME v1 v2 ...
# (schedule switches: bash tools/ab_env.sh DRVAE_TUNE fold_tail=1 fold_tail=5)
name=$1; shift
for v in "$@"; do
env $name=$v python bench.py --steps 3000 --warmup 50 --no-cpu-baseline --no-roofline --no-extras 2>&1 | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); print('$name=$v', r['ms_per_step'], r['losses_last_step']['ELBO'], r['chain_wait_ticks'])"
done
