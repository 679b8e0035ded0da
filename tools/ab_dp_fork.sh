#!/bin/bash
# same-box A/B of the captured-exchange fork (DRVAE_TUNE=dp_fork) over a one-rank RCCL communicator: ms per step and the
# chains' wait counters (site order: join, side start, side mid, optimiser gate, side flag-4/7 wait, noise wait, start park)
export DRVAE_FORCE_DP=1 DRVAE_SIDE_CUS=64
for wl in cfg2 cfg4; do
  for rep in 1; do
    for f in 0 1 2; do
      DRVAE_TUNE=dp_fork=$f python3 bench.py --workload $wl --dp-exchange captured --steps 600 --warmup 30 --no-extras --no-cpu-baseline --no-roofline --no-steady 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$wl dp_fork=$f', d['ms_per_step'], 'ticks/step', [round(t / 630) for t in d['chain_wait_ticks'][1::2]])"
    done
  done
  python3 bench.py --workload $wl --dp-exchange single --steps 600 --warmup 30 --no-extras --no-cpu-baseline --no-roofline --no-steady --no-exchange-modes 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$wl single', d['ms_per_step'], d.get('exchange'))"
done
