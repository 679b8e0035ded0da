"""Input pipeline next to the hot path: the reference's dataset
containers and sampler weights (src/DrVAE.py:880-963, src/VFAE.py:658-721,
src/utils.py:292-327, src/run_drvae.py:150-166), plus a DEVICE-RESIDENT batcher that
feeds the fused step without host round trips.

Difference by design: the reference draws every batch with ``WeightedRandomSampler`` +
``DataLoader(drop_last)``, so the number of rows per data group (labeled/unlabeled x
singleton/pair) fluctuates from batch to batch.  ``DeviceBatcher`` keeps the same per-row
sampling weights but STRATIFIES each batch: a fixed number of rows per group (the expected
composition under the weights), drawn on the GPU, in a fixed group order.  Every batch then
has the same structure, so one step plan and one captured hipGraph serve the whole epoch and
only data (rows, labels) move -- device to device.
"""
import types

import numpy as np
import torch

from . import kernels as K


class DrVAEDataset(torch.utils.data.Dataset):
    """``(x1, x2, s, y, has_x2, has_y)`` rows (src/DrVAE.py:880-905)."""

    FIELDS = ('x1', 'x2', 's', 'y', 'has_x2', 'has_y')

    def __init__(self, x1, x2, s, y, has_x2, has_y):
        n = x1.size(0)
        assert x2.size(0) == n and x1.size(1) == x2.size(1)
        assert y.size(0) == n and s.size(0) == n and has_x2.size(0) == n and has_y.size(0) == n
        self.x1, self.x2, self.s, self.y, self.has_x2, self.has_y = x1, x2, s, y, has_x2, has_y

    def __getitem__(self, index):
        return tuple(getattr(self, f)[index] for f in self.FIELDS)

    def __len__(self):
        return self.x1.size(0)

    def to(self, device):
        """HBM-resident copy (x1/x2 as contiguous fp32: the gather kernels read them in place)."""
        return DrVAEDataset(self.x1.to(device, torch.float32).contiguous(),
                            self.x2.to(device, torch.float32).contiguous(), self.s.to(device), self.y.to(device),
                            self.has_x2.to(device), self.has_y.to(device))


class VFAEDataset(torch.utils.data.Dataset):
    """``(x1, s, y, has_y)`` rows (src/VFAE.py:658-680)."""

    def __init__(self, x1, s, y, has_y):
        n = x1.size(0)
        assert y.size(0) == n and s.size(0) == n and has_y.size(0) == n
        self.x1, self.s, self.y, self.has_y = x1, s, y, has_y

    def __getitem__(self, index):
        return self.x1[index], self.s[index], self.y[index], self.has_y[index]

    def __len__(self):
        return self.x1.size(0)


def wrap_in_DrVAEDataset(sing, pair, y_key='y', concat='both', downlabel_to=None, remove_unlabeled=False):
    """Dict(s) of numpy arrays -> (DrVAEDataset, merged dict)  (src/DrVAE.py:908-963):
    'both' stacks singletons on top of pairs with zero-imputed x2, 'pair_only' / 'sing_only'
    keep one side; ``downlabel_to`` keeps the labels of that many random cell lines (``cid``) and
    marks the rest unlabeled (-66); ``remove_unlabeled`` drops rows without a label."""
    if concat == 'both':
        d = {k: np.concatenate((sing[k], pair[k])) for k in set(sing) & set(pair)}
        d['x2'] = np.concatenate((np.zeros(sing['x1'].shape), pair['x2']))
        d['has_x2'] = np.concatenate((np.zeros(len(sing['x1'])), np.ones(len(pair['x2']))))
    elif concat == 'pair_only':
        d = pair
        d['has_x2'] = np.ones(len(pair['x2']))
    elif concat == 'sing_only':
        d = sing
        d['x2'] = np.zeros(sing['x1'].shape)
        d['has_x2'] = np.zeros(len(sing['x1']))
    else:
        raise ValueError('Invalid parameter for dataset concatenation type')
    if downlabel_to is not None:
        cids = np.unique(d['cid'][d['has_y']])
        np.random.shuffle(cids)
        keep = set(cids[:downlabel_to].tolist())
        drop = np.array([c not in keep for c in d['cid']])
        d['has_y'][drop] = 0
        d['y'][drop] = -66
        d['ycont'][drop] = -66
    if remove_unlabeled:
        sel = d['has_y'] != 0
        for k in list(d):
            d[k] = d[k][sel]
    ds = DrVAEDataset(x1=torch.from_numpy(d['x1']).float(), x2=torch.from_numpy(d['x2']).float(),
                      s=torch.from_numpy(d['s'].astype(np.int32)), y=torch.from_numpy(d[y_key]),
                      has_x2=torch.from_numpy(d['has_x2'].astype(np.int32)),
                      has_y=torch.from_numpy(d['has_y'].astype(np.int32)))
    return ds, d


def compute_balanced_weights(labels, unlabeled_data_ratio=None, unlabeled_token=None):
    """Per-sample weights that make every class equally likely in a minibatch
    (src/utils.py:292-327); with ``unlabeled_data_ratio`` the unlabeled token gets that share."""
    if unlabeled_data_ratio is not None:
        assert unlabeled_token is not None and 0. < unlabeled_data_ratio < 1.
    labels = np.asarray(labels)
    classes, inverse, counts = np.unique(labels, return_inverse=True, return_counts=True)
    w = 1. / counts
    if unlabeled_data_ratio is not None:
        w[classes != unlabeled_token] *= (1. - unlabeled_data_ratio) / (len(classes) - 1)
        w[classes == unlabeled_token] *= unlabeled_data_ratio
    return torch.from_numpy(w[inverse]).double()


_GROUPS = ((1, 0), (0, 0), (1, 1), (0, 1))      # (has_y, has_x2) in the reference's order ls, us, lp, up


class DeviceBatcher:
    """Stratified, weighted, with-replacement minibatches drawn ON the device and written
    straight into the fused step's input buffers (see the module docstring)."""

    def __init__(self, dataset, weights, batch_size, group_counts=None, seed=0, mode='stratified', generator='device',
                 pair_bucket=None, label_bucket=None):
        """``mode='stratified'`` (default): every batch has the SAME composition (the expected counts of the four
        groups under the weights, or ``group_counts``) -- rows within a group drawn with replacement by weight;
        the step runs on the plan of exactly that structure.
        ``mode='sampler'``: the reference's pipeline to the letter (src/run_drvae.py:150-166):
        ``WeightedRandomSampler(weights, len(weights))`` -- len(dataset) i.i.d. draws with replacement by weight --
        cut into consecutive batches with ``drop_last``; the composition of a batch is whatever the draws give, so
        the step runs on the batch-independent ("universal") plan with on-device group masks.
        ``generator='cpu'`` (sampler mode): the epoch's index table is drawn exactly as the reference's pipeline draws
        it -- torch's DEFAULT CPU generator (``torch.manual_seed``): per epoch one int64 draw (the DataLoader iterator's
        base seed) followed by ``torch.multinomial(weights.double(), len(dataset), True)``, full batches kept -- so the
        index stream equals ``list(DataLoader(ds, sampler=WeightedRandomSampler(w, len(w)), drop_last=...))`` bit for
        bit (tests/golden/sampler.npz); the default draws on the device from the batcher's own generator.
        ``pair_bucket`` = w (sampler mode, models with pairs): the batch-independent plan pays for the worst case --
        every row a pair: 3 L B decoder rows.  With buckets the epoch's batches are re-ordered pairs first (the loss is
        a sum over rows: the order inside a batch means nothing) and every batch runs on the plan whose number of pair
        slots is its number of pairs rounded up to a multiple of w -- a handful of plans, each captured once
        (``prepare_epoch`` / ``select``); the decoder then runs L B + 2 L ceil_w(N_pairs) rows.
        ``label_bucket`` = v: the same for the labels.  Rows go in the order unlabeled pairs, labeled pairs, labeled
        singles, unlabeled singles -- the pairs stay a prefix, the labeled rows are ONE run [N_up, N_up + N_l) -- and the
        plan gives the rows [ceil_v(N_up), floor_v(N_up + N_l)), labeled for sure, one fprop row (their class) instead
        of one per class."""
        # a dataset shorter than one batch is ONE batch of all its rows: DataLoader(drop_last=(len >= batch_size))
        batch_size = min(batch_size, len(dataset))
        self.ds, self.batch_size = dataset, batch_size
        self.mode = mode
        assert mode in ('stratified', 'sampler')
        dev = dataset.x1.device
        w = torch.as_tensor(weights, dtype=torch.float64).to(dev)
        hy = dataset.has_y.reshape(-1).bool()
        hx = dataset.has_x2.reshape(-1).bool() if hasattr(dataset, 'has_x2') else torch.zeros_like(hy)
        self.members, self.gweights, mass = [], [], []
        for (gy, gx) in _GROUPS:
            idx = torch.nonzero((hy == bool(gy)) & (hx == bool(gx))).reshape(-1)
            self.members.append(idx)
            self.gweights.append(w[idx].float())
            mass.append(float(w[idx].sum()))
        if group_counts is None:          # expected composition of a batch, largest-remainder rounding
            share = np.asarray(mass) / max(sum(mass), 1e-300) * batch_size
            cnt = np.floor(share).astype(int)
            for j in np.argsort(-(share - cnt))[:batch_size - cnt.sum()]:
                cnt[j] += 1
            group_counts = cnt.tolist()
        assert sum(group_counts) == batch_size and all(c == 0 or len(m) > 0 for c, m in zip(group_counts, self.members))
        self.group_counts = list(group_counts)
        self.has_y = np.concatenate([np.full(c, gy) for c, (gy, gx) in zip(group_counts, _GROUPS)]).astype(np.int64)
        self.has_x2 = np.concatenate([np.full(c, gx) for c, (gy, gx) in zip(group_counts, _GROUPS)]).astype(np.int64)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self._idx32 = torch.zeros(batch_size, dtype=torch.int32, device=dev)
        assert generator in ('device', 'cpu') and (generator == 'device' or mode == 'sampler')
        assert not (pair_bucket or label_bucket) or (mode == 'sampler' and int(pair_bucket or 1) >= 1 and int(label_bucket or 1) >= 1)
        self.pair_bucket = int(pair_bucket) if pair_bucket else None
        self.label_bucket = int(label_bucket) if label_bucket else None
        self.batch_specs = None           # (bucketed) plan of every batch of the current epoch table: (pair slots, a, b)
        self._k = 0                       # (bucketed) batches handed out since begin_epoch
        self.n_switch = 0                 # (bucketed) how many times a step ran on another plan than the one before
        self.cpu_stream = generator == 'cpu'
        self.dp = None                    # (rank, world) under data parallelism: set by ``bind(dp=...)``
        if mode == 'sampler':
            self.weights = w.float()
            self.weights_cpu = torch.as_tensor(weights, dtype=torch.float64).cpu().reshape(-1)
            self.hx32, self.hy32 = hx.to(torch.int32).contiguous(), hy.to(torch.int32).contiguous()
            self._pending = None          # (cpu stream) rows of the current epoch not handed out yet

    @property
    def dataset(self):
        return self.ds

    def __len__(self):
        """batches per epoch: what the reference's DataLoader yields with a len(dataset)-draw
        weighted sampler and drop_last (src/run_drvae.py:150-162)"""
        return max(1, len(self.ds) // self.global_batch)

    @property
    def world(self):
        return self.dp[1] if self.dp else 1

    @property
    def rank(self):
        return self.dp[0] if self.dp else 0

    @property
    def global_batch(self):
        """rows of one optimiser step over all ranks: ``batch_size`` is PER RANK (weak scaling)"""
        return self.batch_size * self.world

    def _n_tot(self):
        return self.global_batch if self.dp else None

    def bind(self, engine, counts=None, dp=None):
        """build / select the step plan for this batcher's fixed batch structure.  ``dp`` = (rank, world): data
        parallelism -- ``batch_size`` rows per rank, a global batch of world x batch_size rows per step.
        Every rank holds the same dataset, weights and seed and therefore draws the SAME global index table; it runs
        columns [rank B, (rank + 1) B) of it.  The loss normalisers are the global batch's: N_total = world B, and
        N_pairs / N_labeled of every batch are counted over the global table (``gcounts``, read by the feed launch) --
        no communication.  The Philox draws are keyed by the row's position in the global batch (``engine.row0``)."""
        dp = None if dp is None else (int(dp[0]), int(dp[1]))
        sig = (id(engine), dp, counts, self.pair_bucket, self.label_bucket)
        if getattr(self, '_bound', None) == sig and engine.plan is getattr(self, '_bound_plan', None) and not self.bucketed:
            if dp is not None:
                engine.row0 = dp[0] * self.batch_size
            return engine.plan          # (every epoch of ``fit`` binds: nothing to rebuild)
        self.engine = engine
        self.dp = dp
        assert self.dp is None or 0 <= self.dp[0] < self.dp[1]
        if self.dp is not None:
            assert not engine.cfg.use_MMD, 'use_MMD: the MMD penalty is a cross-row term, it cannot be sharded over ranks'
            engine.row0 = self.rank * self.batch_size
        if engine.cfg.use_s and (engine.cfg.use_MMD or self.mode == 'sampler'):
            raise NotImplementedError('DeviceBatcher: models conditioned on the nuisance variable (use_s extension) run on '
                                      'stratified device batches without the MMD penalty (its row lists are host knowledge); '
                                      'otherwise feed them through run_on_batch / a tuple loader')
        if self.mode == 'sampler':
            assert counts is None, 'sampler feed: the global counts of every batch come from the shared index table (dp=...)'
            engine.universal = True
            if self.pair_bucket and not engine.cfg.has_pert:
                self.pair_bucket = None                      # (no pairs in this model: nothing to bucket)
            if self.label_bucket and not engine.cfg.has_y:
                self.label_bucket = None
            self._bound, self._bound_plan = sig, engine.set_structure_universal(self.batch_size, n_tot=self._n_tot())
            return self._bound_plan
        if counts is None and self.dp is not None:      # every rank's batch has the same composition
            cfg = engine.cfg
            n_lab = int(self.has_y.sum()) if cfg.has_y else 0
            n_tot = n_lab if (cfg.kind == 'vfae' and not cfg.semi_supervised) else self.batch_size
            counts = tuple(self.world * c for c in (n_tot, int(self.has_x2.sum()) if cfg.has_pert else 0, n_lab))
        engine.set_structure(self.has_x2, self.has_y, counts)
        self._bound, self._bound_plan = sig, engine.plan
        return engine.plan

    def begin_epoch(self, n_batches=None, table=None):
        """Draw the index table of a whole epoch on the device (one multinomial per group) and
        install it as the bound plan's graph-resident feed: from here on every captured train step
        gathers its own minibatch (``dv_batch_feed``), i.e. an epoch is ``len(self)`` graph replays
        with no other host work.  Returns the table (n_batches, batch_size) int32 -- under data parallelism this
        rank's columns of the global table.  ``table``: the GLOBAL table (n_batches, world x batch_size) given instead of
        drawn (tests; replaying a recorded epoch)."""
        eng, p = self.engine, self.engine.plan
        if eng.cfg.use_s:
            raise NotImplementedError('use_s: the graph-resident epoch feed does not carry the nuisance classes; use feed()')
        n_b = len(self) if n_batches is None else n_batches
        fd = p.feed
        want_gc = self.dp is not None and self.mode == 'sampler'
        if fd is None or fd.owner is not self or fd.n_batches != n_b or (getattr(fd, 'gcounts', None) is not None) != want_gc:
            dev = self.ds.x1.device
            fd = types.SimpleNamespace(owner=self, n_batches=n_b, x1=self._feed_rows(self.ds.x1),
                                       x2=self._feed_rows(getattr(self.ds, 'x2', None)) if eng.cfg.has_pert else None,
                                       y32=None if eng.cfg.cont else self.ds.y.reshape(-1).to(torch.int32).contiguous(),
                                       yf=self.ds.y.reshape(len(self.ds), -1).float().contiguous() if eng.cfg.cont else None,
                                       table=torch.zeros(n_b, self.batch_size, dtype=torch.int32, device=dev),
                                       base=torch.zeros(1, dtype=torch.int32, device=dev),
                                       gcounts=torch.zeros(n_b, 2, dtype=torch.int32, device=dev)
                                       if (self.dp is not None and self.mode == 'sampler') else None)
            if self.mode == 'sampler':
                fd.hx32, fd.hy32 = self.hx32, self.hy32
                if fd.y32 is None:
                    fd.y32 = self.hy32
            p.feed = fd
        else:
            self._refresh_feed_rows(fd)
        p.feed_active = True
        if self.mode == 'sampler':
            # WeightedRandomSampler: i.i.d. draws over ALL rows; DataLoader(drop_last): consecutive full batches
            if table is not None:
                draws = torch.as_tensor(table).to(fd.table.device)
                assert tuple(draws.shape) == (n_b, self.global_batch)
            elif self._take_ahead(n_b) is not None:
                draws = self._ahead_draws
            elif self.cpu_stream:
                rows = []
                while sum(len(r) for r in rows) < n_b:
                    rows.append(self._reference_epoch())
                draws = torch.cat(rows)[:n_b].to(fd.table.device)
            else:
                draws = torch.multinomial(self.weights, n_b * self.global_batch, replacement=True, generator=self.gen)
            tab = draws.reshape(n_b, self.global_batch)
            self.global_table = tab
            if self.dp is not None:
                # the batch's normalisers are those of the GLOBAL batch: counted here over the shared table, read by the
                # step's first launch (dv_batch_feed: masks.gcounts) -- no collective, no host round trip
                g = tab.long()
                zero = torch.zeros(n_b, dtype=torch.int64, device=tab.device)
                fd.gcounts.copy_(torch.stack([self.hx32[g].sum(1) if eng.cfg.has_pert else zero,
                                              self.hy32[g].sum(1) if eng.cfg.has_y else zero], 1))
                tab = tab[:, self.rank * self.batch_size:(self.rank + 1) * self.batch_size]
            if self.bucketed:
                self._order_and_specs(tab)
                tab = self._tab_sorted
            fd.table.copy_(tab)
            fd.base.copy_(eng.step_dev)
            if self.bucketed:                    # the same table feeds whichever of the bucket plans a batch runs on
                self._feed = fd
                for spec in sorted(set(self.batch_specs) | {self._full_spec}):
                    q = self._plan(spec)
                    q.feed, q.feed_active = fd, True
                self._plan(self.batch_specs[0])
            return fd.table
        if table is not None:
            tab = torch.as_tensor(table).to(fd.table.device).reshape(n_b, self.world, self.batch_size)[:, self.rank]
        elif self._take_ahead(n_b) is not None:
            tab = self._ahead_draws
        else:
            # (data parallelism: world x c draws per group and batch from the shared generator; this rank's c of them)
            R, r = self.world, self.rank
            parts = [m[torch.multinomial(w, c * R * n_b, replacement=True, generator=self.gen)].reshape(n_b, R, c)[:, r]
                     for m, w, c in zip(self.members, self.gweights, self.group_counts) if c > 0]
            tab = torch.cat(parts, 1)
        fd.table.copy_(tab)
        fd.base.copy_(eng.step_dev)         # device to device: batch index = optimiser step - base
        return fd.table

    def draw_ahead(self, n_batches=None):
        """Draw the NEXT epoch's index table now (``fit`` calls this once the current epoch's replays are enqueued): the
        dozen small launches of the draw then queue behind the running steps instead of standing between two epochs with
        the GPU idle.  The batcher's generator is consumed in the same order as
        without it -- the tables are the same.  Not with ``generator='cpu'``: that stream is torch's DEFAULT generator,
        whose order against the caller's other draws is the reference's."""
        if self.cpu_stream or self.engine.cfg.use_s:
            return
        n_b = len(self) if n_batches is None else n_batches
        dev = self.ds.x1.device
        ev = None
        if dev.type == 'cuda':
            # on a stream of its own: the draw's small launches run NEXT TO the epoch's steps, not behind them
            if getattr(self, '_draw_stream', None) is None:
                self._draw_stream = torch.cuda.Stream(device=dev)
            ctx = torch.cuda.stream(self._draw_stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            if self.mode == 'sampler':
                d = torch.multinomial(self.weights, n_b * self.global_batch, replacement=True, generator=self.gen)
            else:
                R, r = self.world, self.rank
                d = torch.cat([m[torch.multinomial(w, c * R * n_b, replacement=True, generator=self.gen)].reshape(n_b, R, c)[:, r]
                               for m, w, c in zip(self.members, self.gweights, self.group_counts) if c > 0], 1)
            if dev.type == 'cuda':
                ev = torch.cuda.Event()
                ev.record()
        self._ahead = (n_b, self.mode, self.dp, d, ev)

    def _take_ahead(self, n_b):
        """the table drawn ahead for this epoch, if there is one for this number of batches / mode / dp state"""
        a, self._ahead = getattr(self, '_ahead', None), None
        self._ahead_draws = None
        if a is not None and a[:3] == (n_b, self.mode, self.dp):
            if a[4] is not None:
                torch.cuda.current_stream().wait_event(a[4])       # (drawn on the batcher's own stream)
                a[3].record_stream(torch.cuda.current_stream())
            self._ahead_draws = a[3]
        return self._ahead_draws

    def rebase(self):
        """the current epoch table again from its first batch: batch index = optimiser step - base (after replays that
        were not training steps of the epoch: CU-split tuning, a capture's warm-up).  No draw: under data parallelism
        every rank must consume the shared generator alike, whatever it captures or tunes"""
        fd = self.engine.plan.feed
        fd.base.copy_(self.engine.step_dev)
        self._k = 0

    def _refresh_feed_rows(self, fd):
        """the padded copies of ``_feed_rows`` follow a dataset edited IN PLACE between two epochs (normalisation,
        augmentation): checked by the tensors' version counters at every ``begin_epoch``, refreshed in place -- the captured
        steps keep reading the same buffer (round-4 advisor: train and evaluation must see the same data)"""
        for name in ('x1', 'x2'):
            x, cur = getattr(self.ds, name, None), getattr(fd, name, None)
            if x is None or cur is None or cur.data_ptr() == x.data_ptr():
                continue
            seen = self.__dict__.setdefault('_rows_version', {})
            if seen.get(name) != (x.data_ptr(), x._version):
                cur.copy_(x)
                seen[name] = (x.data_ptr(), x._version)

    def _feed_rows(self, x):
        """the dataset as the graph-resident feed reads it: fp32 rows 16-B aligned -- a copy with padded rows when the gene
        count is no multiple of 4 (978): the feed's row gather is then 16-B loads instead of 4-B ones (epoch feed 0.1932 ->
        0.1916 ms per step).  Kept current by ``_refresh_feed_rows`` when the dataset is edited in place"""
        if x is None or x.device.type != 'cuda':
            return x
        if x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
            return x
        if x.dim() != 2 or x.numel() * 4 > (2 << 30):
            return x
        cache = self.__dict__.setdefault('_rows_cache', {})
        key = (x.data_ptr(), tuple(x.shape), x._version)
        if key not in cache:
            buf = torch.zeros(x.shape[0], (x.shape[1] + 3) // 4 * 4, dtype=torch.float32, device=x.device)
            buf[:, :x.shape[1]].copy_(x)
            if len(cache) > 4:
                cache.clear()
            cache[key] = buf[:, :x.shape[1]]
        return cache[key]

    @property
    def bucketed(self):
        return bool(self.pair_bucket or self.label_bucket)

    @property
    def batch_slots(self):
        """pair slots of every batch of the current epoch table"""
        return np.asarray([sp[0] for sp in self.batch_specs], np.int64)

    @property
    def _full_spec(self):
        return (self.batch_size, 0, 0)

    def _plan(self, spec):
        p = self.engine.set_structure_universal(self.batch_size, spec[0], spec[1:], n_tot=self._n_tot())
        self.engine.pinned_plans.add(p.key)       # (the plan cache is small and drops what is not captured)
        return p

    def _order_and_specs(self, tab):
        """re-order every batch of the epoch table (stable inside a group) and choose its plan: the epoch's ONE host sync"""
        B, eng = self.batch_size, self.engine
        hx = self.hx32[tab.long()] if eng.cfg.has_pert else torch.zeros_like(tab)
        hy = self.hy32[tab.long()] if eng.cfg.has_y else torch.zeros_like(tab)
        if self.label_bucket:        # unlabeled pairs | labeled pairs | labeled singles | unlabeled singles
            grp = torch.where(hx != 0, hy, 3 - hy)
        else:                        # pairs | singles
            grp = 1 - hx
        self._tab_sorted = tab.gather(1, torch.argsort(grp, dim=1, stable=True))
        cnt = torch.stack([hx.sum(1), (hx * (1 - hy)).sum(1), hy.sum(1)], 1).cpu().numpy().astype(np.int64)
        specs = []
        for n_p, n_up, n_l in cnt:
            P = B
            if self.pair_bucket:
                w = self.pair_bucket
                P = int(np.clip(-(-n_p // w) * w, min(w, B), B))
            a = b = 0
            if self.label_bucket:
                v = self.label_bucket
                a, b = int(min(-(-n_up // v) * v, B)), int((n_up + n_l) // v * v)
                if b <= a:
                    a = b = 0
            specs.append((P, a, b))
        self.batch_specs = specs
        self._k = 0

    def prepare_epoch(self, capture):
        """(bucketed sampler feed) capture the plans this epoch's batches run on that are not captured yet --
        ``capture(engine)`` is called with each selected (a handful per run, ~0.1 s each, in the first epochs only) --
        and the every-row-may-be-anything plan, which serves any batch"""
        if not self.bucketed:
            return
        eng = self.engine
        for spec in sorted(set(self.batch_specs) | {self._full_spec}):
            p = self._plan(spec)
            if not eng.use_capture(p.key):
                assert p.live_feed is self._feed
                capture(eng)
                eng.stash_capture()
        self._ready = {sp for sp in getattr(self, '_ready', set()) | set(self.batch_specs) | {self._full_spec}
                       if eng.use_capture(self._plan(sp).key)}
        self.select(0)

    def select(self, k=None):
        """(bucketed sampler feed) make the plan of the epoch's batch ``k`` (default: the next one) current -- its own
        if captured, else the cheapest captured one that serves it (at least its pairs' slots, a labeled range inside
        its labeled run); the caller replays right after.  Without buckets: nothing to do."""
        if not self.bucketed:
            return
        if k is None:
            k = self._k
        self._k = k + 1
        eng = self.engine
        want = self.batch_specs[min(k, len(self.batch_specs) - 1)]
        cur = eng.plan
        if want not in getattr(self, '_ready', ()) or not eng.use_capture(self._plan(want).key):
            # (a composition first met after the captures were made, e.g. a re-drawn table)
            ok = [sp for sp in getattr(self, '_ready', ())
                  if sp[0] >= want[0] and (sp[1] == sp[2] or (sp[1] >= want[1] and sp[2] <= want[2]))]
            ok.sort(key=lambda sp: 2 * sp[0] - (sp[2] - sp[1]))
            if not (ok and eng.use_capture(self._plan(ok[0]).key)):
                raise RuntimeError('DeviceBatcher: no captured step for this feed (call prepare_epoch after begin_epoch; '
                                   'a table of another size is a new feed: its steps are captured again)')
        self.n_switch += eng.plan is not cur

    def _reference_epoch(self):
        """(len(self), batch_size) int64 on the host: one epoch of the reference's loader, drawn from torch's default
        CPU generator in the reference's order (src/run_drvae.py:150-162 on torch.utils.data): the DataLoader iterator
        takes its base seed first, the sampler then draws len(dataset) rows at once; the incomplete last batch is
        dropped"""
        torch.empty((), dtype=torch.int64).random_()
        draws = torch.multinomial(self.weights_cpu, len(self.weights_cpu), True)
        n_b = len(self)
        return draws[:n_b * self.global_batch].reshape(n_b, self.global_batch)

    def next_indices(self):
        """the next batch's dataset rows -- under data parallelism the GLOBAL batch (world x batch_size rows, the same
        on every rank); ``feed`` takes this rank's share"""
        if self.mode == 'sampler' and self.cpu_stream:
            if self._pending is None or len(self._pending) == 0:
                self._pending = self._reference_epoch()
            idx, self._pending = self._pending[0], self._pending[1:]
            return idx.to(self.ds.x1.device)
        if self.mode == 'sampler':
            return torch.multinomial(self.weights, self.global_batch, replacement=True, generator=self.gen)
        R = self.world
        parts = [m[torch.multinomial(w, c * R, replacement=True, generator=self.gen)].reshape(R, c)
                 for m, w, c in zip(self.members, self.gweights, self.group_counts) if c > 0]
        return torch.cat(parts, 1).reshape(-1)

    def feed(self, idx=None):
        """draw the next batch and write it into the bound engine's buffers (device to device)"""
        if self.mode == 'sampler' and self.bucketed:
            # explicit batches come in the order they were drawn: the plan without assumptions about the row order
            self.engine.set_structure_universal(self.batch_size, n_tot=self._n_tot())
        p = self.engine.plan
        p.feed_active = False       # this batch is explicit data in XSRC, not a row of the epoch table
        idx = self.next_indices() if idx is None else idx
        if self.dp is not None and idx.numel() == self.global_batch:
            if self.mode == 'sampler':      # the global batch's (N_pairs, N_labeled): data of dv_batch_masks
                zero = torch.zeros((), dtype=torch.int32, device=idx.device)
                p.gcounts_dev.copy_(torch.stack([self.hx32[idx].sum() if self.engine.cfg.has_pert else zero,
                                                 self.hy32[idx].sum() if self.engine.cfg.has_y else zero]))
            idx = idx.reshape(self.world, self.batch_size)[self.rank]
        self._idx32.copy_(idx)
        K.rows_gather(p.XSRC[:p.B], self.ds.x1, self._idx32)
        if self.engine.cfg.has_pert:
            K.rows_gather(p.XSRC[p.B:], self.ds.x2, self._idx32)
        if self.engine.cfg.use_s:           # one-hot(s) columns of the encoder / decoder inputs (device to device)
            p.set_s_device(self.ds.s.reshape(-1)[idx])
        if self.mode == 'sampler':          # group membership of THIS batch: data for dv_batch_masks
            if self.engine.cfg.has_pert:
                p.hx_dev.copy_(self.hx32[idx])
            if self.engine.cfg.has_y:
                p.hy_dev.copy_(self.hy32[idx])
                p.y_dev.copy_(self.ds.y.reshape(-1)[idx].to(torch.int32))
            return idx
        if self.engine.cfg.has_y and self.engine.cfg.cont:
            p.ylab.copy_(self.ds.y.reshape(len(self.ds), -1)[idx].float())
        elif self.engine.cfg.has_y:
            p.set_labels_device(self.ds.y.reshape(-1)[idx])
        return idx
