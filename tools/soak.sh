# long replays of every schedule variant: finite losses, no device-wait time-outs (even words of chain_wait_ticks)
run() { python bench.py --steps 30000 --warmup 50 --no-cpu-baseline --no-roofline --no-extras "$@" 2>&1 | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); w = r['chain_wait_ticks'] or []
print('%-34s %.4f ms/step finite=%s wait_errors=%s' % ('$*', r['ms_per_step'], r['finite'], [v for v in w[0::2] if v]))"; }
run --workload cfg2
run --workload cfg4
run --workload cfg1
run --workload cfg2 --feed epoch
run --workload cfg2 --feed sampler
