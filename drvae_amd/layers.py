"""Drop-in for the reference's ``layers`` module (``import layers as lyr``,
src/blocks.py:12): same class names and constructor signatures, parameters named and
shaped as in the reference ``state_dict`` -- the arithmetic runs in the HIP kernels.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops


class WeightNormLinear(nn.Linear):
    """Linear layer with weight normalisation, ``out = (g/||W_row||) * (x W^T) + b``
    (src/layers.py:8-41).  Parameters: ``weight (out,in)``, ``bias (out)``, ``g (out)``.

    The per-call ``g.ne(g).all()`` NaN probe of src/layers.py:27 (a host sync) is
    replaced by a host-side flag; data-dependent init runs once, on the first forward,
    when ``data_init=True`` (never requested by the reference's own callers).
    """

    def __init__(self, in_features, out_features, data_init=False, init_scale=1.):
        super().__init__(in_features, out_features, bias=True)
        fill = float('nan') if data_init else 1.0
        self.g = nn.Parameter(torch.full((out_features,), fill))
        self.init_scale = init_scale
        self._pending_data_init = bool(data_init)

    @torch.no_grad()
    def _data_init(self, x):
        # intended semantics of src/layers.py:27-35 (its expand_as is shape-buggy unless in==out)
        self.weight.normal_().mul_(0.05)
        wn = self.weight / torch.norm(self.weight, 2, 1, keepdim=True)
        out = ops.linear_act([x.float().contiguous()], wn.contiguous(), None)      # (the product on the MFMA GEMM: dv_gemm)
        scale = self.init_scale / torch.sqrt(out.var(0) + 1e-10)
        self.g.copy_(scale)
        self.bias.copy_(-out.mean(0) * scale)
        self._pending_data_init = False

    def forward(self, x):
        if self._pending_data_init:
            self._data_init(x)
        return ops.linear_act([x], self.weight, self.bias, self.g)


def _made_mask(n_in, n_out, m_pre, output_layer, rev_order):
    """Connectivity mask of a MADE layer (Germain et al. 2015): hidden unit j may see
    input i iff m_pre[i] <= m[j]; output unit j iff m_pre[i] < m[j]."""
    m_pre = np.asarray(m_pre)
    if output_layer:
        m = np.arange(1, n_out + 1)
        if rev_order:
            m = m[::-1]
        mask = (m_pre[:, None] < m[None, :])
    else:
        base = np.arange(1, int(np.max(m_pre))).astype(int)
        m = np.resize(base, n_out)          # cyclic repeat of 1..D-1, truncated to n_out units
        mask = (m_pre[:, None] <= m[None, :])
    return m, mask.astype('float32')


class MaskedLinear(nn.Linear):
    """MADE layer container (src/layers.py:44-139).  As in the reference, the masked
    ``forward`` is unreachable (it is defined after a ``return`` inside the mask builder,
    src/layers.py:135-139), so the layer acts as a plain Linear; ``mask``, ``m`` and
    ``get_m()`` are provided for API parity."""

    def __init__(self, in_features, out_features, m_pre, output_layer, rev_order=False):
        total_in = int(np.asarray(in_features).sum())
        super().__init__(total_in, out_features, bias=True)
        self.output_layer, self.rev_order = output_layer, rev_order
        if isinstance(in_features, (tuple, list)):
            x_dim, h_dim = in_features[0], int(np.asarray(in_features[1:]).sum())
        else:
            x_dim, h_dim = in_features, 0
        if m_pre is None:
            m_pre = np.arange(1, x_dim + 1).astype(int)
            if rev_order:
                m_pre = m_pre[::-1]
            if h_dim > 0:
                m_pre = np.concatenate((m_pre, np.ones(h_dim, dtype=int)))
        self.m_pre = m_pre
        self.m, mask = _made_mask(x_dim + h_dim, out_features, m_pre, output_layer, rev_order)
        self.mask = torch.from_numpy(mask)

    def get_m(self):
        return self.m

    def forward(self, x):
        return ops.linear_act([x], self.weight, self.bias)
