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


def test_headline_roofline_follows_from_the_line_and_carries_every_workload():
    """round-4 review, item 2: ONE definition -- ``roofline.frac`` = algorithmic GFLOP per step / this line's
    ``ms_per_step`` / the fp32-MFMA peak -- recomputable from the line; the isolated / in-graph / steady figures keep
    keys of their own; and the other configurations (cfg 5 above all) ride inside ``roofline`` with the same definition"""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--strict']
    env = dict(os.environ)
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    rf = r['roofline']
    assert rf['bound'] == 'mfma' and rf['peak'] == 157.3 and rf['ms_per_step'] == r['ms_per_step']
    want = rf['algorithmic_gflop_per_step'] / r['ms_per_step'] / rf['peak']
    assert rf['frac'] == pytest.approx(want, rel=2e-3)
    assert rf['achieved'] == pytest.approx(want * rf['peak'], rel=2e-3)
    assert 'step_level' not in rf and rf['isolated']['frac'] > 0 and rf['steady']['ms_per_step'] == r['steady_state']['ms_per_step']
    for name in ('cfg1', 'cfg4', 'cfg5'):
        w = rf['workloads'][name]
        assert w['frac'] == pytest.approx(w['algorithmic_gflop_per_step'] / w['ms_per_step'] / rf['peak'], rel=2e-3)
        assert w['ms_per_step'] == r['other_workloads'][name]['ms_per_step'] and w['steps'] > 0
    assert rf['workloads']['cfg5']['frac'] > 0.5          # the one configuration where the MFMA roofline binds
    assert r['finite'] and not r.get('extras_failed')
