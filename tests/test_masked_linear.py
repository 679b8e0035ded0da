"""CPU: ``layers.MaskedLinear`` against the reference's own MADE masks
(tests/golden/masked_linear.npz, generated from /root/reference/src/layers.py:44-133 by
tests/golden/make_golden.py).  Integer / 0-1 work: the bar is bit-exact."""
import numpy as np
import pytest
import torch

from tests.golden import cases as C


@pytest.mark.parametrize('tag', list(C.masked_linear_cases()))
def test_masked_linear_matches_reference_bit_exactly(tag):
    from drvae_amd import layers as lyr
    gold = C.load('masked_linear')
    m_pre = None
    for li, (in_f, out_f, output_layer, rev) in enumerate(C.masked_linear_cases()[tag]):
        lay = lyr.MaskedLinear(in_f, out_f, m_pre, output_layer, rev_order=rev)
        mask, m = gold['%s/%d/mask' % (tag, li)], gold['%s/%d/m' % (tag, li)]
        assert isinstance(lay, torch.nn.Linear) and tuple(lay.weight.shape) == tuple(gold['%s/%d/weight_shape' % (tag, li)])
        assert lay.mask.dtype == torch.float32 and tuple(lay.mask.shape) == mask.shape
        assert np.array_equal(lay.mask.numpy(), mask)                       # (d_in_total, d_out), as the reference
        assert np.array_equal(np.asarray(lay.get_m()).astype(np.int64), m)
        assert np.array_equal(np.asarray(lay.m).astype(np.int64), m)
        assert np.array_equal(np.asarray(lay.m_pre).astype(np.int64), gold['%s/%d/m_pre' % (tag, li)])
        assert lay.output_layer == output_layer and lay.rev_order == rev
        m_pre = lay.get_m()


def test_masked_linear_has_no_masked_forward_like_the_reference():
    """the reference's masked ``forward`` sits after a ``return`` inside the mask builder (src/layers.py:135-139):
    the class inherits nn.Linear.forward, i.e. the mask is NOT applied.  Ours documents and keeps that."""
    from drvae_amd import layers as lyr
    lay = lyr.MaskedLinear(4, 3, None, False)
    assert 'unreachable' in lyr.MaskedLinear.__doc__
    assert float(lay.mask.min()) == 0.0        # there is something the forward pass could have masked
