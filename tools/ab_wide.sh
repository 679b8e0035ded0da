# A/B of schedule switches over the wide configuration (cfg 5): bash tools/ab_wide.sh "nll_cs=0,early_adam=0" "nll_cs=1,early_adam=0" ...
for e in "$@"; do
env DRVAE_TUNE=$e python bench.py --workload wide --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --no-steady 2>&1 | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); print('$e', r['ms_per_step'], r['losses_last_step']['ELBO'])"
done
