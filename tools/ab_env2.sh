# bash tools/ab_env2.sh "A=1 B=2" "A=3" ...: one bench run per quoted environment set
for e in "$@"; do
env $e python bench.py --steps 3000 --warmup 50 --no-cpu-baseline --no-roofline --no-extras 2>&1 | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); print('$e', r['ms_per_step'], r['config'].get('side_chain_cus'), r['chain_wait_ticks'])"
done
