# the measurement table of DESIGN.md in one go (GPU box): one bench line per configuration
run() { python bench.py --steps 2000 --warmup 50 --no-cpu-baseline --no-roofline --no-extras "$@" 2>&1 | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); print('%-40s %.4f ms/step  %10.0f samples/s  %s' % ('$*', r['ms_per_step'], r['value'], r['config'].get('side_chain_cus')))"; }
run --workload cfg2
run --workload cfg4
run --workload cfg1
run --workload cfg2 --feed epoch
run --workload cfg2 --feed sampler
DRVAE_FORCE_DP=1 MASTER_PORT=29571 run --workload cfg2
