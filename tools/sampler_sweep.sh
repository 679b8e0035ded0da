for e in "$@"; do
env $e python bench.py --steps 2000 --warmup 50 --no-cpu-baseline --no-roofline --feed sampler 2>&1 | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); print('$e', r['ms_per_step'], r['config'].get('side_chain_cus'), r['chain_wait_ticks'])"
done
