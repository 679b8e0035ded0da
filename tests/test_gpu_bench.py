"""bench.py's graph-resident feeds on the GPU box (round-3 advisor findings): the bucketed sampler feed replays every
batch on ITS plan, and the steady region re-draws the index table in place (the captured feed object stays current)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=600):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-roofline', '--no-extras', '--strict'] + list(args)
    env = dict(os.environ)
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])


def test_sampler_feed_replays_each_batch_on_its_plan():
    # more steps than one index table holds (1024 batches): the table is re-drawn in place inside the timed loop
    r = _bench('--feed', 'sampler', '--steps', '1100', '--warmup', '5', '--no-steady', '--check-feed')
    assert r['finite'] and r['steps'] == 1100


@pytest.mark.parametrize('feed_args', [['--feed', 'epoch'], ['--feed', 'sampler', '--pair-bucket', '0', '--label-bucket', '0']])
def test_graph_resident_feeds_survive_the_steady_region(feed_args):
    # the steady region (default on) used to ask for a table of another size: a NEW feed object, and the next
    # replay failed the captured-feed assertion
    r = _bench(*feed_args, '--steps', '40', '--warmup', '5')
    assert r['finite'] and r['steady_state']['steps'] >= 200
