"""Flat fp32 parameter arena: every parameter of a model is a view into ONE contiguous
device buffer, with sibling buffers for gradients and the Adam moments.

Why: (1) the optimiser is a single fused kernel over the arena (HBM-bound, 7 words per
parameter) instead of ~30 tiny per-tensor updates; (2) the data-parallel gradient
exchange is an RCCL all-reduce of ``xchg`` = [loss scalars | gradients] (in two pieces when
overlapped with backward); (3) the two heads of every Gaussian block are laid out back to back so that one
GEMM with a split epilogue evaluates both.  Names, shapes and (out,in)
row-major layout of the reference ``state_dict`` are preserved: the nn.Parameters
simply alias the arena.  A weight matrix whose inner dimension is no multiple of 4 (W1: 800 x 978, the
fprop blocks' 200 x 102) is stored with its rows padded to 16 B -- the nn.Parameter is the
``[:, :in]`` view of it, the pad columns are zero and stay zero (their gradient is never written, and
Adam / Adamax with L2 leave a zero parameter with a zero gradient where it is) -- so that every
product of the step has 16-B aligned operand rows (the LDS-DMA GEMM kernels need them).
"""
from collections import OrderedDict

import torch

N_LOSS = 8   # RECL KLD PERT YL MMD ELBO CMPL + 1 spare, parked in FRONT of the gradients

_HEAD_PAIRS = (('encoder_mu.linear_mu', 'encoder_lv.linear_lv'),
               ('encoder_mu.linear_mu', 'encoder_sg.linear_sg'))


def _fusion_groups(names, frozen=()):
    """order parameter names so that fused-head partners are adjacent: returns list of groups
    (``frozen`` names are never fused: they live behind the optimised range)"""
    names = [n for n in names if n not in frozen]
    used, groups = set(), []
    for n in names:
        if n in used:
            continue
        grp = [n]
        for a, b in _HEAD_PAIRS:
            for suffix in ('.weight', '.bias', '.g'):
                if n.endswith(a + suffix):
                    partner = n[:-len(a + suffix)] + b + suffix
                    if partner in names:
                        grp.append(partner)
        if n.endswith('.W_mu'):      # DiagGaussianModuleLinear: W_mu | encoder_lv.linear_lv.weight
            partner = n[:-len('W_mu')] + 'encoder_lv.linear_lv.weight'
            if partner in names:
                grp.append(partner)
        if n.endswith('.bias_mu'):
            partner = n[:-len('bias_mu')] + 'encoder_lv.linear_lv.bias'
            if partner in names:
                grp.append(partner)
        used.update(grp)
        groups.append(grp)
    return groups


def row_stride(shape):
    """floats between consecutive rows of a parameter of this shape inside the arena"""
    if len(shape) == 2 and shape[1] >= 16 and shape[1] % 4:
        return (shape[1] + 3) // 4 * 4
    return shape[-1] if len(shape) else 1


def span(t):
    """floats from the first to one past the last element of a (row-padded) arena view"""
    if t.dim() == 2 and t.numel():
        return (t.shape[0] - 1) * t.stride(0) + t.shape[1]
    return t.numel()


class ParamArena:
    def __init__(self, named_shapes, device, frozen=()):
        """named_shapes: OrderedDict name -> shape (reference state_dict names).  ``frozen``: parameters
        that never receive a gradient; torch's optimisers skip such parameters altogether (no weight
        decay), so they are parked behind the range the fused optimiser kernel sweeps (``n_live``)."""
        self.device = torch.device(device)
        self.shapes = OrderedDict((k, tuple(v)) for k, v in named_shapes.items())
        self.offsets = {}
        off = 0
        for grp in _fusion_groups(self.shapes, frozen) + [[n] for n in self.shapes if n in frozen]:
            if grp[0] in frozen and not hasattr(self, 'n_live'):
                self.n_live = (off + 3) // 4 * 4
            off = (off + 3) // 4 * 4                       # 16-B aligned group start
            for n in grp:
                self.offsets[n] = off
                off += self.numel(n)
        self.n_params = (off + 3) // 4 * 4
        if not hasattr(self, 'n_live'):
            self.n_live = self.n_params
        PAD = 16    # tail slack: GEMM edge tiles may over-read a row end by up to 3 floats (dv_gemm_desc.flags)
        z = lambda n: torch.zeros(n + PAD, dtype=torch.float32, device=self.device)[:n]
        self.param = z(self.n_params)
        # one exchange buffer [loss scalars | gradients]; the decoder_x block (the largest, and the first
        # whose gradients are final in the backward pass) is the tail, so the data-parallel exchange can
        # send it early: ``xchg[late_end:]`` while the rest of backward runs, ``xchg[:late_end]`` after
        self.xchg = z(N_LOSS + self.n_params)
        self.loss = self.xchg[:N_LOSS]
        self.grad = self.xchg[N_LOSS:]
        self.exp_avg = z(self.n_params)
        self.exp_avg_sq = z(self.n_params)
        early = [o for n, o in self.offsets.items() if n.startswith('decoder_x.')]
        names = list(self.offsets)
        first = min(early) if early else self.n_params
        # valid only when decoder_x really is the tail of the arena
        tail_ok = early and all(o >= first for n, o in self.offsets.items() if n.startswith('decoder_x.')) and \
            all(o < first for n, o in self.offsets.items() if not n.startswith('decoder_x.') and n not in frozen)
        self.late_end = N_LOSS + (first if tail_ok else self.n_params)

    def numel(self, name):
        """floats the parameter occupies in the arena (rows padded to 16 B: see ``row_stride``)"""
        s = self.shapes[name]
        if len(s) == 2:
            return s[0] * row_stride(s)
        n = 1
        for d in s:
            n *= d
        return n

    def _view(self, buf, name):
        o, s = self.offsets[name], self.shapes[name]
        if len(s) == 2 and row_stride(s) != s[1]:
            return buf[o:o + self.numel(name)].view(s[0], row_stride(s))[:, :s[1]]
        return buf[o:o + self.numel(name)].view(s)

    def pads(self, buf):
        """the pad columns of every row-padded parameter inside ``buf`` (param / grad / a moment): zero, always"""
        out = []
        for name, s in self.shapes.items():
            if len(s) == 2 and row_stride(s) != s[1]:
                o = self.offsets[name]
                out.append(buf[o:o + self.numel(name)].view(s[0], row_stride(s))[:, s[1]:])
        return out

    def p(self, name):
        return self._view(self.param, name)

    def g(self, name):
        return self._view(self.grad, name)

    def fused(self, buf, first, second):
        """(N1+N2, ...) view covering two adjacent parameters (the two heads of a block)."""
        o = self.offsets[first]
        assert self.offsets[second] == o + self.numel(first), (first, second)
        s1, s2 = self.shapes[first], self.shapes[second]
        assert s1[1:] == s2[1:]
        shape = (s1[0] + s2[0],) + s1[1:]
        if len(s1) == 2 and row_stride(s1) != s1[1]:
            return buf[o:o + self.numel(first) + self.numel(second)].view(shape[0], row_stride(s1))[:, :s1[1]]
        return buf[o:o + self.numel(first) + self.numel(second)].view(shape)

    def adopt(self, module):
        """Make every parameter of ``module`` an alias of the arena (values are copied in;
        ``.grad`` aliases the gradient arena).  ``module`` must already live on the device."""
        sd_names = [k for k, _ in module.named_parameters()]
        assert set(sd_names) == set(self.shapes), set(sd_names) ^ set(self.shapes)
        with torch.no_grad():
            for name, prm in module.named_parameters():
                v = self.p(name)
                v.copy_(prm.data.to(self.device))
                prm.data = v
                prm.grad = self.g(name)
        return self

    def state_dict(self):
        return OrderedDict((k, self.p(k)) for k in self.shapes)

    def load(self, named_arrays):
        with torch.no_grad():
            for k, a in named_arrays.items():
                self.p(k).copy_(torch.as_tensor(a, dtype=torch.float32).to(self.device))
