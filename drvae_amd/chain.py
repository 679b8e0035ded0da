"""Linear layers bound to arena views and chains of them (trunk + head of one block) evaluated on stacked
rows with pre-allocated buffers: the GEMM-level building block of the fused train step."""
import torch

from . import kernels as K


def _pad4(n):
    return (n + 3) // 4 * 4


def _whole_rows(t):
    """is ``t`` made of whole rows of a row-padded 2-D buffer (``zeros(r, pad4(c))[:, :c]`` or a row range of one)?"""
    b = t._base
    return (b is not None and b.dim() == 2 and t.dim() == 2 and b.is_contiguous() and t.stride() == (b.shape[1], 1)
            and b.shape[1] == _pad4(t.shape[1]) and (t.storage_offset() - b.storage_offset()) % b.shape[1] == 0)


class _Lin:
    """One Linear layer (or two fused heads) bound to arena views."""

    def __init__(self, arena, wname, bname, gname=None, second=None, act='identity', act1=None, shift0=0.0,
                 shift1=0.0):
        if second is None:
            self.W, self.dW = arena.p(wname), arena.g(wname)
            self.b, self.db = arena.p(bname), arena.g(bname)
            self.g = arena.p(gname) if gname else None
            self.dg = arena.g(gname) if gname else None
            self.split = self.W.shape[0]
        else:
            w2, b2, g2 = second
            self.W, self.dW = arena.fused(arena.param, wname, w2), arena.fused(arena.grad, wname, w2)
            self.b, self.db = arena.fused(arena.param, bname, b2), arena.fused(arena.grad, bname, b2)
            self.g = arena.fused(arena.param, gname, g2) if gname and g2 else None
            self.dg = arena.fused(arena.grad, gname, g2) if gname and g2 else None
            self.split = arena.shapes[wname][0]
        self.N, self.Kin = self.W.shape
        self.act0, self.act1 = act, (act if act1 is None else act1)
        self.shift0, self.shift1 = shift0, shift1
        dev = self.W.device
        if self.g is not None:
            self.scale = torch.empty(self.N, device=dev)
            self.norm = torch.empty(self.N, device=dev)
            self.raw = torch.empty(self.N, self.Kin, device=dev)
        else:
            self.scale = self.norm = self.raw = None


class _Chain:
    """trunk layers + final layer of one block evaluated on M stacked rows, with buffers.

    Inputs handed to ``forward`` whose width is no multiple of 4 must either have unpadded rows (row stride == width)
    or be WHOLE row-padded buffers with zero pads (``plan.mat``): the products then run over the padded K (``kpad`` of
    ``kernels._gemm_desc``; the arena's weights are row-padded the same way).  ``backward`` likewise runs the weight
    gradient of a single-source layer and the data gradients into its own buffers (or into a destination made of whole
    padded rows) over the padded N (``npad``): the pad columns receive the zeros they hold anyway."""

    def __init__(self, layers, M, device, resid_cols=0):
        self.layers, self.M, self.resid_cols = layers, M, resid_cols
        self.out = [torch.zeros(M, _pad4(l.N), device=device)[:, :l.N] for l in layers]
        # gradient w.r.t. the pre-activation of every layer but the last (the caller owns that one)
        self.dpre = [torch.zeros(M, _pad4(l.N), device=device)[:, :l.N] for l in layers[:-1]]

    @staticmethod
    def _pad_ok(t):
        """may a product over ``t`` run over its padded width?  Only when the caller's contract is checkable: rows that are
        not padded at all, or WHOLE rows of a row-padded buffer (``plan.mat`` / this chain's own buffers / the arena's
        padded weights), whose pad columns are zero by construction.  A column slice of a wider buffer (``Q[:, :Z]`` with
        ld = 2 Z) has neighbours, not zeros, behind its last column (round-4 advisor)."""
        if t.dim() != 2 or t.shape[1] % 4 == 0:
            return True
        ld = t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
        return ld == t.shape[1] or _whole_rows(t)

    def raw_last_ok(self):
        """may the last layer run as a plain product (no bias, no activation), finished by its consumer?"""
        l = self.layers[-1]
        return l.g is None and l.split * 2 == l.N and l.act0 == 'identity' and self.resid_cols == 0

    def forward(self, inputs, resid=None, publish=None, heads=None, raw_last=False):
        """``publish`` = (flag, counter, add): the FIRST launch of the chain publishes on entry.
        ``raw_last``: the last layer's launch is the plain product x W^T -- bias, the second head's activation and its
        shift are left to the consumer (``K.nll_rows_fwdbwd(bias=...)``: chip-filling heads, plain GEMM epilogue).
        ``heads`` = dict(sample=...) | dict(nll=..., out=...): the dual-head last layer runs as ``K.linear_heads``
        with that row work fused into its epilogue (with ``nll`` the heads themselves are NOT stored: ``out``
        receives their gradients)."""
        x = list(inputs)
        for li, l in enumerate(self.layers):
            if l.g is not None:
                K.wn_scale(l.scale, l.norm, l.W, l.g)
            last = li == len(self.layers) - 1
            kpad = len(x) == 1 and self._pad_ok(x[0])      # (only inputs whose pad columns are zero BY CONSTRUCTION)
            if last and heads is not None:
                assert l.split * 2 == l.N
                K.linear_heads(heads.get('out', self.out[li]), x[0], l.W, l.b, split=l.split,
                               x2=x[1] if len(x) > 1 else None, scale=l.scale, act0=l.act0, act1=l.act1,
                               shift0=l.shift0, shift1=l.shift1, resid=resid,
                               resid_cols=self.resid_cols if resid is not None else 0, overread=True,
                               publish=publish if (li == 0 and l.g is None) else None,
                               sample=heads.get('sample'), nll=heads.get('nll'), kpad=kpad)
                return self.out[-1]
            if last and raw_last:
                assert self.raw_last_ok() and resid is None
                K.gemm(self.out[li], x[0], l.W, True, True, A2=x[1] if len(x) > 1 else None, overread=True,
                       publish=publish if li == 0 else None, kpad=kpad)
                return self.out[-1]
            K.linear_fwd(self.out[li], x[0], l.W, l.b, x2=x[1] if len(x) > 1 else None, scale=l.scale, split=l.split,
                         act0=l.act0, act1=l.act1, shift0=l.shift0, shift1=l.shift1,
                         resid=resid if last else None, resid_cols=self.resid_cols if (last and resid is not None) else 0,
                         overread=True, publish=publish if (li == 0 and l.g is None) else None, kpad=kpad)
            x = [self.out[li]]
        return self.out[-1]

    def backward(self, dpre_last, inputs, dinputs=None, wbranch=None, publish_after_last=None, publish_first=None,
                 db_last_done=False, klq=None):
        """dpre_last: gradient w.r.t. the last layer's pre-activation.  ``dinputs``: per input
        source a list of (dst, alpha, beta) destinations for its gradient (or None to skip).
        ``wbranch``: optional side stream for the weight-gradient GEMMs (they are leaves: only
        Adam reads them), so that they overlap the dx chain.  ``publish_after_last`` = (flag, counter, add):
        the first launch AFTER the last layer's launches publishes the flag on entry (= both gradients of the
        last layer are final and its weights are no longer read); ``publish_first``: the chain's FIRST launch
        does (= everything in front of this backward pass is complete).  ``db_last_done``: the
        last layer's bias gradient has been written by the producer of ``dpre_last`` already (``kernels.nll_rows_raw_cs``).
        ``klq`` (see ``kernels.linear_bwd_pair``): the chain's input is a sample of q rows -- the FIRST layer's data-gradient
        launch writes d/d(mu | logvar) of those rows instead of d/d(input); returns True when the launch took it (a paired
        first layer), else the caller runs the row pass itself."""
        dpre = dpre_last
        pending_pub = publish_first
        took_klq = False
        n_layers = len(self.layers)
        for li in range(n_layers - 1, -1, -1):
            l = self.layers[li]
            db = None if (db_last_done and li == n_layers - 1) else l.db
            srcs = list(inputs) if li == 0 else [self.out[li - 1]]
            if li == len(self.layers) - 2 and publish_after_last is not None:
                pending_pub = publish_after_last

            def wgrad(l=l, srcs=srcs, dpre=dpre, db=db):
                dW = l.raw if l.g is not None else l.dW
                c0 = 0
                for si, s in enumerate(srcs):
                    w = s.shape[1]
                    K.linear_bwd_weight(dW[:, c0:c0 + w], dpre, s, dbias=db if si == 0 else None, overread=True,
                                        npad=len(srcs) == 1 and l.g is None and self._pad_ok(s))
                    c0 += w
                if l.g is not None:
                    K.wn_bwd(l.dW, l.dg, l.raw, l.W, l.g, l.norm)

            # the layer's weight- and data-gradient both only need dpre: one paired launch when the
            # layer has a single input source, no WeightNorm and a single data-gradient destination
            single_dst = li > 0 or (dinputs is not None and len(srcs) == 1 and dinputs[0] is not None
                                    and len(dinputs[0]) == 1)
            if wbranch is None and l.g is None and len(srcs) == 1 and single_dst:
                if li > 0:
                    prev = self.layers[li - 1]
                    K.linear_bwd_pair(l.dW, db, self.dpre[li - 1], dpre, srcs[0], l.W, yref=self.out[li - 1],
                                      act=prev.act0, shift=prev.shift0, overread=True, publish=pending_pub,
                                      npad=self._pad_ok(srcs[0]), npad_x=True)
                    dpre = self.dpre[li - 1]
                else:
                    dst, alpha, beta = dinputs[0][0]
                    if klq is not None and beta == 0.0:
                        K.linear_bwd_pair(l.dW, db, None, dpre, srcs[0], l.W, alpha=alpha, overread=True, publish=pending_pub,
                                          npad=self._pad_ok(srcs[0]), klq=klq)
                        took_klq = True
                    else:
                        K.linear_bwd_pair(l.dW, db, dst, dpre, srcs[0], l.W, alpha=alpha, beta_x=beta, overread=True,
                                          publish=pending_pub, npad=self._pad_ok(srcs[0]), npad_x=_whole_rows(dst))
                pending_pub = None
                continue
            if pending_pub is not None:      # (no paired launch for this layer: a launch of its own)
                K.flag_publish(*pending_pub)
                pending_pub = None
            if wbranch is not None:
                with wbranch:
                    wgrad()
            else:
                wgrad()
            if li > 0:
                prev = self.layers[li - 1]
                K.linear_bwd_data(self.dpre[li - 1], dpre, l.W, kscale=l.scale, yref=self.out[li - 1], act=prev.act0,
                                  shift=prev.shift0, overread=True, npad=True)
                dpre = self.dpre[li - 1]
            elif dinputs is not None:
                c0 = 0
                for si, s in enumerate(srcs):
                    w = s.shape[1]
                    for (dst, alpha, beta) in (dinputs[si] or []):
                        K.linear_bwd_data(dst, dpre, l.W[:, c0:c0 + w], kscale=l.scale, alpha=alpha, beta=beta,
                                          overread=True)
                    c0 += w
        if wbranch is not None:
            wbranch.join()
        return took_klq
