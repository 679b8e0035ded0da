"""Build-container only (needs /root/reference): ONE full run of tests/golden/make_golden.py reproduces every
committed fixture bit for bit -- the provenance of tests/golden/*.npz is that script and nothing else."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir('/root/reference/src'), reason='the reference is only mounted in the build container')
def test_one_run_of_make_golden_reproduces_every_fixture(tmp_path):
    env = dict(os.environ, DRVAE_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'golden', 'make_golden.py')], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    committed = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', '*.npz')))
    fresh = sorted(glob.glob(os.path.join(str(tmp_path), '*.npz')))
    assert [os.path.basename(f) for f in committed] == [os.path.basename(f) for f in fresh]
    for a, b in zip(committed, fresh):
        ga, gb = np.load(a), np.load(b)
        assert sorted(ga.files) == sorted(gb.files), os.path.basename(a)
        for k in ga.files:
            assert ga[k].dtype == gb[k].dtype and ga[k].shape == gb[k].shape, (os.path.basename(a), k)
            assert np.array_equal(ga[k], gb[k], equal_nan=ga[k].dtype.kind == 'f'), (os.path.basename(a), k)
