"""The flat parameter arena: reference names and shapes as views, fused heads adjacent, rows padded to 16 B."""
from collections import OrderedDict

import torch

from drvae_amd.arena import ParamArena, row_stride, span


def _arena():
    shapes = OrderedDict([('enc.l0.weight', (5, 18)), ('enc.l0.bias', (5,)),
                          ('enc.encoder_mu.linear_mu.weight', (3, 22)), ('enc.encoder_mu.linear_mu.bias', (3,)),
                          ('enc.encoder_lv.linear_lv.weight', (4, 22)), ('enc.encoder_lv.linear_lv.bias', (4,)),
                          ('clf.weight', (2, 7)), ('dec.weight', (6, 16))])
    return ParamArena(shapes, 'cpu'), shapes


def test_rows_padded_to_16_bytes_views_keep_reference_shapes():
    a, shapes = _arena()
    assert row_stride((5, 18)) == 20 and row_stride((6, 16)) == 16 and row_stride((2, 7)) == 7 and row_stride((5,)) == 5
    for k, s in shapes.items():
        v = a.p(k)
        assert tuple(v.shape) == s and tuple(a.g(k).shape) == s
        assert a.offsets[k] % 4 == 0 or k.endswith('lv.weight') or k.endswith('lv.bias')
        if len(s) == 2:
            assert v.stride() == (row_stride(s), 1)
    assert span(a.p('enc.l0.weight')) == 4 * 20 + 18 and span(a.p('enc.l0.bias')) == 5
    # values go into the live columns only; the pads stay zero
    a.load({k: torch.full(s, 2.0) for k, s in shapes.items()})
    assert float(a.param.sum()) == 2.0 * sum(torch.Size(s).numel() for s in shapes.values())
    assert len(a.pads(a.param)) == 3 and not any(p.any() for p in a.pads(a.param))
    sd = a.state_dict()
    assert all(tuple(sd[k].shape) == s and bool((sd[k] == 2).all()) for k, s in shapes.items())


def test_fused_heads_share_the_padded_row_stride():
    a, _ = _arena()
    W = a.fused(a.param, 'enc.encoder_mu.linear_mu.weight', 'enc.encoder_lv.linear_lv.weight')
    assert tuple(W.shape) == (7, 22) and W.stride() == (24, 1)
    a.p('enc.encoder_mu.linear_mu.weight').fill_(1.0)
    a.p('enc.encoder_lv.linear_lv.weight').fill_(3.0)
    assert bool((W[:3] == 1).all()) and bool((W[3:] == 3).all())
    b = a.fused(a.param, 'enc.encoder_mu.linear_mu.bias', 'enc.encoder_lv.linear_lv.bias')
    assert tuple(b.shape) == (7,)


def test_adopted_module_parameters_alias_the_arena():
    lin = torch.nn.Linear(18, 5)
    a = ParamArena(OrderedDict((k, tuple(v.shape)) for k, v in lin.named_parameters()), 'cpu')
    w0 = lin.weight.detach().clone()
    a.adopt(lin)
    assert torch.equal(lin.weight.detach(), w0) and lin.weight.data_ptr() == a.param.data_ptr() + 4 * a.offsets['weight']
    y = lin(torch.ones(2, 18)).sum()
    y.backward()
    assert torch.equal(a.g('weight'), torch.full((5, 18), 2.0)) and not any(p.any() for p in a.pads(a.grad))
