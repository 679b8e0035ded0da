"""The ``fit`` protocol of the three models: epoch loop, whole-set evaluation,
rolling-mean early stopping and the snapshot policy of the reference
(src/DrVAE.py:743-877, src/PVAE.py:554-672, src/VFAE.py:523-656), on top of the fused step.

Two ways to feed the epoch loop:
  * a ``drvae_amd.data.DeviceBatcher`` -- the dataset lives in HBM, batches are drawn on the
    device, and every step is ONE hipGraph replay (no host<->device traffic, one host sync
    per epoch for the logged train loss);
  * any iterable of batch tuples (e.g. the reference's ``torch.utils.data.DataLoader``):
    compatibility path, one ``run_on_batch`` per tuple.
"""
import time
from collections import OrderedDict

import numpy as np
import torch

from . import engine as E
from . import metrics as MET


class EarlyStopping:
    """The patience / rolling-mean / snapshot controller spelled out inline in the reference's
    ``fit`` (src/DrVAE.py:755-766, 829-868).  One ``update`` per epoch."""

    def __init__(self, epochs, early_stop, patience=50, patience_increase=15, improvement_threshold=0.999,
                 memory_length=3, save_model_after=0):
        self.epochs, self.early_stop = epochs, early_stop
        self.patience, self.patience_increase = patience, patience_increase
        self.improvement_threshold = improvement_threshold
        self.memory_length, self.save_model_after = memory_length, save_model_after
        self.best = -np.inf
        self.rolling = np.array([], dtype=float)
        self.since_improvement = 0
        self.snapshotted = False

    def update(self, epoch, valid_obj):
        """-> dict(snapshot, stop, patience_hit, continuing, rolling_mean, best_before)"""
        self.since_improvement += 1
        self.rolling = np.append(self.rolling, valid_obj)[-self.memory_length:]
        out = dict(best_before=self.best, rolling=self.rolling.copy(), snapshot=False, stop=False,
                   patience_hit=False, continuing=False)
        score = out['rolling_mean'] = self.rolling.mean()
        if score * self.improvement_threshold > self.best:      # False for nan: no improvement
            self.patience = max(self.patience, epoch + self.patience_increase)
            self.best = score
            self.since_improvement = 0
        if (self.early_stop and self.since_improvement == self.save_model_after) or \
                (self.patience <= epoch and not self.snapshotted):
            self.snapshotted = out['snapshot'] = True
        if self.patience <= epoch:
            out['patience_hit'] = True
            if self.early_stop:
                if not self.snapshotted:
                    self.snapshotted = out['snapshot'] = True
                out['stop'] = True
            else:                                               # keep training to the last epoch
                out['continuing'] = True
                self.patience = self.epochs + 1
                self.snapshotted = False
        out['best'] = self.best
        return out

    def on_interrupt(self):
        """KeyboardInterrupt: snapshot unless one was already taken (src/DrVAE.py:869-874)"""
        take = not self.snapshotted
        self.snapshotted = True
        return take


_REC = 'RMSE: {:.3f} R2: {:.3f} Pearson: {:.3f}'
_NAN4 = ('rmse', 'r2', 'pearr', 'll')


class FitMixin:
    """evaluate_performance / fit for ``ELBOModel`` (``self.kind`` selects the model's variant)."""
    fit_patience = 50         # src/DrVAE.py:756, src/PVAE.py:567 (VFAE: 40, src/VFAE.py:536)

    # ------------------------------------------------------------------ metrics
    def eval_y_prediction(self, pred, proba, ylab):
        if getattr(self, 'type_y', 'discrete') != 'discrete':
            return MET.eval_y_regression(pred, ylab)
        return MET.eval_y_prediction(pred, proba, ylab, self.dim_y)

    def _np(self, t):
        return t.detach().cpu().numpy()

    @torch.no_grad()
    def _evaluate(self, x1, x2, s, y, has_x2, has_y, return_full_data=False):
        """one pass of losses + means-only inference + metrics over a row set
        (src/DrVAE.py:640-741, src/PVAE.py:479-552, src/VFAE.py:472-521)"""
        kind = self.kind
        dev = next(self.parameters()).device
        dv = lambda t: None if t is None else t.to(dev)
        x1, x2, y, has_x2, has_y = dv(x1), dv(x2), dv(y), dv(has_x2), dv(has_y)
        perf = OrderedDict()
        kw = dict(x1=x1, s=s)
        if kind != 'vfae':
            kw.update(x2=x2, has_x2=has_x2)
        if kind != 'pvae':
            kw.update(y=y, has_y=has_y)
        try:
            losses = self.run_on_batch(train_mode=False, **kw)
            perf['losses'] = OrderedDict((k, v.clone()) for k, v in losses.items())
        except Exception as e:          # the reference evaluates on regardless (src/DrVAE.py:648-654)
            if kind == 'pvae':
                raise
            print('Warning, computation of losses failed in evaluation!')
            print(e)
            perf['losses'] = None
        res = self.forward(x1, s)
        parts = []
        if kind != 'pvae':
            yidx = torch.nonzero(has_y.reshape(-1)).reshape(-1)
            cont = getattr(self, 'type_y', 'discrete') != 'discrete'
            ylab = y.reshape(y.shape[0], -1)[yidx] if cont else y.reshape(-1)[yidx]
            for k, v in self.eval_y_prediction(res['pred'][yidx], res['proba'][yidx], ylab).items():
                perf['y_' + k] = v
            if cont:
                parts.append('Y: RMSE: {:.3f} R2: {:.3f} Pearson: {:.3f}'.format(perf['y_rmse'], perf['y_r2'],
                                                                                 perf['y_pearr']))
            else:
                parts.append('Y: Accuracy: {:.3f}% AUROC: {:.3f} AUPR: {:.3f}'.format(
                    perf['y_acc'] * 100., perf['y_auroc'], perf['y_aupr']))
        for k, v in self.eval_x_reconstruction(x1, *res['px1']).items():
            perf['x1_' + k] = v
        parts.append('X1: ' + _REC.format(perf['x1_rmse'], perf['x1_r2'], perf['x1_pearr']))
        x2idx = None
        if kind != 'vfae':
            x2idx = torch.nonzero(has_x2.reshape(-1)).reshape(-1)
            if len(x2idx) > 0:
                x2p = x2[x2idx]
                rp = self.eval_x_reconstruction(x2p, *[t_[x2idx] for t_ in res['px2'][:2]])
                for k, v in rp.items():
                    perf['x2_' + k] = v
                parts.append('X2: ' + _REC.format(perf['x2_rmse'], perf['x2_r2'], perf['x2_pearr']))
            else:
                for k in _NAN4:
                    perf['x2_' + k] = np.nan
                parts.append('X2: no x2 data')
        if return_full_data:
            perf['z1'] = self._np(res['z1'])
            if kind != 'vfae':
                perf['z2'] = self._np(res['z2'])
                perf['x2_pert'] = self._np(res['x2_pert'])
            if kind != 'pvae':
                perf['pred'] = self._np(res['pred'])
                perf['proba'] = self._np(res['proba'])
            if kind != 'vfae':
                self._evaluate_identity_pert(perf, res, x1, x2, s, x2idx, yidx if kind == 'drvae' else None,
                                             ylab if kind == 'drvae' else None)
        perf['model_class'] = self.__class__.__name__
        return perf, '\t '.join(parts)

    def _evaluate_identity_pert(self, perf, res, x1, x2, s, x2idx, yidx, ylab):
        """the extra report with the perturbation function set to the identity
        (src/DrVAE.py:696-738, src/PVAE.py:514-549)"""
        res2 = self.forward_w_pert_identity(x1, x2, s)
        if yidx is not None:
            for k, v in self.eval_y_prediction(res2['pred'][yidx], res2['proba'][yidx], ylab).items():
                perf['y_wI_' + k] = v
        if len(x2idx) > 0:
            x2p = x2[x2idx]
            for tag, key in (('x2_wI_', 'px2'), ('x2_rec_', 'px2_rec')):
                rp = self.eval_x_reconstruction(x2p, *[t_[x2idx] for t_ in res2[key][:2]])
                for k, v in rp.items():
                    perf[tag + k] = v
            q1, q2 = res2['z1'][x2idx].double(), res2['z2'][x2idx].double()
            pz = res['z2'][x2idx].double()
            perf['qz1mu_qz2mu_rmse'] = float(torch.sqrt(((q1 - q2) ** 2).mean()))
            perf['pz2Fz1mu_qz2mu_rmse'] = float(torch.sqrt(((pz - q2) ** 2).mean()))
            qz1 = [t[x2idx].contiguous() for t in res2['qz1']]
            qz2 = [t[x2idx].contiguous() for t in res2['qz2']]
            pz2 = [t[x2idx].contiguous() for t in res['pz2']]
            perf['KL_qz2_qz1'] = float(self.encoder_z1.kldivergence_perx(*(qz2 + qz1)).mean())
            perf['KL_qz2_pz2Fz1'] = float(self.encoder_z1.kldivergence_perx(*(qz2 + pz2)).mean())
        else:
            for tag in ('x2_wI_', 'x2_rec_'):
                for k in _NAN4:
                    perf[tag + k] = np.nan
            for k in ('qz1mu_qz2mu_rmse', 'pz2Fz1mu_qz2mu_rmse', 'KL_qz2_qz1', 'KL_qz2_pz2Fz1'):
                perf[k] = np.nan

    def evaluate_performance_on_dataset(self, ds, return_full_data=False):
        """Whole-set evaluation (src/DrVAE.py:797,821 call it on the train and on the validation set every epoch).  For
        an HBM-resident dataset it is ONE hipGraph replay and ONE device->host copy: eval-mode losses, means-only
        inference, reconstruction statistics and the prediction metrics captured once per (model, dataset) -- see
        ``_EvalGraph``.  ``return_full_data`` (arrays for the caller), ``use_s`` models, continuous targets and host
        datasets take the step-by-step path."""
        g = lambda k: getattr(ds, k, None)
        ev = None if return_full_data else _EvalGraph.get(self, ds)
        if ev is not None:
            return ev.run()
        return self._evaluate(g('x1'), g('x2'), g('s'), g('y'), g('has_x2'), g('has_y'), return_full_data)

    # ------------------------------------------------------------- the objective
    def _valid_objective(self, perf):
        """what early stopping maximises (src/DrVAE.py:822-825, src/PVAE.py:617, src/VFAE.py:600-603)"""
        if self.kind == 'pvae':
            return perf['x1_pearr'] + perf['x2_pearr']
        if getattr(self, 'type_y', 'discrete') != 'discrete':         # src/DrVAE.py:824-825
            v = perf['y_r2'] + perf['y_pearr'] + perf['x1_pearr']
        else:
            v = perf['y_auroc'] + perf['y_aupr'] + perf['x1_pearr']
        return v + perf['x2_pearr'] if self.kind == 'drvae' else v

    def _batch_kwargs(self, batch):
        if self.kind == 'vfae':
            x1, s, y, has_y = batch
            return dict(x1=x1, s=s, y=y, has_y=has_y)
        x1, x2, s, y, has_x2, has_y = batch
        if self.kind == 'pvae':
            return dict(x1=x1, x2=x2, s=s, has_x2=has_x2)
        return dict(x1=x1, x2=x2, s=s, y=y, has_x2=has_x2, has_y=has_y)

    def _train_objective(self, loss):
        v = loss['RECL']
        return v + self.yloss_rate * loss['YL'] if 'YL' in loss else v

    def _log_losses(self, epoch, seen, n_data, frac, loss):
        keys = [k for k in ('CMPL', 'ELBO', 'RECL', 'PERT', 'YL') if k in loss]
        txt = '\t'.join('{}: {:.3f}'.format(k, float(loss[k])) for k in keys)
        self.w2log('training epoch: {} [{}/{} ({:.0f}%)]\t{}'.format(epoch, seen, n_data, 100. * frac, txt))

    # --------------------------------------------------------------- epoch loops
    def _epoch_device(self, batcher, epoch, verbose):
        """one epoch of hipGraph replays fed by the device batcher; returns the mean train objective"""
        eng = self.engine()
        self._assert_arena_aliased()
        batcher.bind(eng, counts=getattr(self, '_global_counts', None), dp=self._dp)
        eng.add_noise = bool(self.add_noise)
        eng.iters = self.finished_training_iters
        resident_feed = not eng.cfg.use_s      # (the nuisance classes travel with host-driven gathers: feed())
        if resident_feed:
            batcher.begin_epoch()           # this epoch's index table; the graph gathers batch b itself
            if getattr(batcher, 'bucketed', False):
                eng.use_capture(eng.plan.key)      # (one captured step per number-of-pairs bucket: this one's, if any)
        else:
            batcher.feed()
        if getattr(eng, '_graph_key', None) != eng.plan.key or getattr(eng, '_graph_noise', None) != eng.add_noise \
                or getattr(eng, '_graph_feed', None) is not eng.plan.live_feed:
            eng.capture(split_for_allreduce=self._allreduce is not None)
            eng._graph_noise = eng.add_noise
            bucketed = getattr(batcher, 'bucketed', False)
            if bucketed:
                eng.stash_capture()         # (under this plan's key: begin_epoch below selects another)
            if self._allreduce is None and not (bucketed and getattr(eng, '_side_cus', None)):
                eng.tune_partition()        # CU split of the two launch chains, by timing (state restored)
            if resident_feed:
                batcher.rebase()            # (the tuning replays advanced the step counter: re-base the table)
        if getattr(batcher, 'bucketed', False):         # a plan per number-of-pairs bucket: capture the missing ones
            def cap(e):
                e.capture(split_for_allreduce=self._allreduce is not None)
                e._graph_noise = e.add_noise
            batcher.prepare_epoch(cap)
        with eng.partition():
            return self._epoch_device_body(eng, batcher, epoch, verbose)

    def _epoch_device_body(self, eng, batcher, epoch, verbose):
        n_b = len(batcher)
        every = max(10, n_b / 10)
        total = torch.zeros((), device=eng.dev)
        # single process: the launch that assembles a step's loss scalars also adds them to ``eng.loss_sum`` -- nothing
        # sits between two replays on the critical chain's stream (three tiny torch launches per step did: 54 steps x
        # ~40 us per epoch at cfg 2).  Under data parallelism the step's scalars are the exchanged ones: summed here
        in_graph = self._allreduce is None
        side = eng.flag_side if getattr(eng, '_side_graph', None) is not None else None
        if in_graph:
            eng.loss_sum.zero_()
            if side is not None:           # (the side chain's stream runs the assembling launch)
                side.wait_stream(torch.cuda.current_stream())
        for b in range(n_b):
            if eng.plan.live_feed is None:
                batcher.feed()
            else:
                batcher.select(b)              # (bucketed sampler feed: this batch's plan; otherwise nothing)
            eng.replay(allreduce=self._allreduce)
            if not in_graph:
                total += self._train_objective(self._loss_tensors(eng))
            if verbose and b % every == 0:
                self._log_losses(epoch, b * batcher.batch_size, len(batcher.dataset), b / n_b, self._loss_tensors(eng))
        self.finished_training_iters = eng.iters
        if eng.plan.live_feed is not None and epoch < self.epochs:
            batcher.draw_ahead()           # (the next epoch's table: queued behind the running steps)
        if in_graph:
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            sums = OrderedDict((k, eng.loss_sum[E.LOSS_IDX[k]]) for k in self._loss_tensors(eng))
            total = self._train_objective(sums)
        # the epoch's ONE host sync: the objective and the sticky words of the device-side waits travel in one copy (a second
        # wait left the device idle once more per epoch)
        eng.join_side()
        if eng.dev.type == 'cuda' and torch.is_tensor(total):
            both = torch.stack([total.reshape(()).double(), eng.sync_err[0::2].abs().sum().double()]).cpu().tolist()
            if both[1] != 0:
                eng.check_sync()           # (raises with the wait sites)
            return both[0] / n_b
        mean = float(total) / n_b
        eng.check_sync()
        return mean

    def _graph_step(self, kw):
        """One train step of a tuple-loader batch through ONE captured graph whatever the batch's mix of pairs / labels
        (the reference's own input pipeline yields a different mix every batch, src/run_drvae.py:150-162): the batch goes
        into the buffers of the batch-independent plan (``set_batch``: device-to-device copies + its flags), the
        captured step turns the flags into group masks itself (dv_batch_masks).  The first batch of a size runs
        eagerly (iteration 0 carries beta_pert = 0.01) and the capture follows."""
        eng = self._batch_to_engine(dp=True, **kw)
        eng.add_noise = bool(self.add_noise)
        eng.iters = self.finished_training_iters
        fresh = getattr(eng, '_graph_key', None) != eng.plan.key or getattr(eng, '_graph_noise', None) != eng.add_noise \
            or getattr(eng, '_graph_feed', None) is not eng.plan.live_feed
        if fresh and eng.iters == 0:
            eng.train_step(allreduce=self._allreduce)
        else:
            if fresh:
                eng.capture(split_for_allreduce=self._allreduce is not None)
                eng._graph_noise = eng.add_noise
            with eng.partition():
                eng.replay(allreduce=self._allreduce)
        self.finished_training_iters = eng.iters
        return self._loss_tensors(eng)

    def _epoch_loader(self, loader, epoch, verbose):
        self._assert_arena_aliased()
        n_b = len(loader)
        every = max(10, n_b / 10)
        total = None
        eng = self.engine()
        # tuple loaders: one batch-independent plan + one captured graph for every batch composition (opt out:
        # ``model.universal_plan = False`` -> a plan per structure, eager launches)
        mode = getattr(self, 'universal_plan', True)       # True | False | 'eager' (the universal plan, eager launches)
        graph = bool(mode) and eng.universal_ok() and eng.dev.type == 'cuda' and \
            getattr(self, '_global_counts', None) is None
        if graph:
            eng.universal = True
            graph = mode != 'eager'
        for b, batch in enumerate(loader):
            self.train()
            loss = self._graph_step(self._batch_kwargs(batch)) if graph else \
                self.run_on_batch(train_mode=True, **self._batch_kwargs(batch))
            v = self._train_objective(loss)
            total = v.clone() if total is None else total + v
            if verbose and b % every == 0:
                self._log_losses(epoch, b * len(batch[0]), len(loader.dataset), b / n_b, loss)
        return float(total) / n_b

    def fit(self, train_loader, valid_loader, add_noise=False, verbose=False, early_stop=False,
            model_filename='best_model.pth'):
        """Train for ``self.epochs`` epochs with the reference's validation / early-stopping /
        snapshot protocol (signature and log lines of src/DrVAE.py:743)."""
        from .data import DeviceBatcher
        np.set_printoptions(precision=4)
        ctl = EarlyStopping(self.epochs, early_stop, patience=self.fit_patience)
        self.w2log('Starting training at: {}'.format(time.strftime('%c')))
        epoch = 0
        try:
            self.add_noise = add_noise
            for epoch in range(1, self.epochs + 1):
                t = time.time()
                if isinstance(train_loader, DeviceBatcher):
                    train_loss = self._epoch_device(train_loader, epoch, verbose)
                else:
                    train_loss = self._epoch_loader(train_loader, epoch, verbose)
                train_perf, train_str = self.evaluate_performance_on_dataset(train_loader.dataset)
                self.w2log('====> Epoch: {}\tIter: {}'.format(epoch, self.finished_training_iters))
                self.w2log('Train: sec/epoch: {:.2f}\tAvg train loss: {:9.4f}\t{}'.format(time.time() - t, train_loss,
                                                                                     train_str))
                t = time.time()
                valid_perf, perf_str = self.evaluate_performance_on_dataset(valid_loader.dataset)
                valid_loss = self._valid_objective(valid_perf)
                self.w2log('Valid: sec/epoch: {:.2f}\tValid set loss: {:9.4f}\t{}'.format(time.time() - t, valid_loss,
                                                                                      perf_str))
                d = ctl.update(epoch, valid_loss)
                self.w2log('Valid rolling mem: {}\tmean: {:.4f}\tbest: {:.4f}'.format(d['rolling'], d['rolling_mean'],
                                                                                    d['best_before']))
                if d['snapshot']:
                    if self.dp_rank == 0:        # (data parallelism: the replicas are identical; one writer)
                        self.save_to_file(model_filename)
                    self.w2log('* Snapshotting at epoch {}'.format(epoch))
                if d['patience_hit']:
                    self.w2log('Early stopping at: {} with train: {:.4f} valid: {:.4f} evaluate_valid_obj: {:.4f} '
                               'best_valid_obj: {:.4f}'.format(epoch, train_loss, valid_loss, d['rolling_mean'],
                                                               d['best']))
                    if d['stop']:
                        break
                    self.w2log('Continuing')
        except KeyboardInterrupt:
            self.w2log('KeyboardInterrupt')
            if ctl.on_interrupt() and self.dp_rank == 0:
                self.save_to_file(model_filename)
                self.w2log('* Snapshotting at epoch {}'.format(epoch))
        self.w2log('Finished training at: {}'.format(time.strftime('%c')))
        self._fit_controller = ctl
        return


class _EvalGraph:
    """The whole-set evaluation of one dataset as a captured launch sequence (round 4).

    What ``_evaluate`` does step by step -- with a host round trip behind nearly every number: index lists from
    ``torch.nonzero``, numpy combinations of the reconstruction partials, ``float()`` of every metric -- is recorded
    here ONCE into a hipGraph over the dataset's own HBM-resident tensors:
      * the evaluation-mode loss scalars of the fused step's forward (Philox draws from the device counter) on a plan of
        the whole set, its inputs gathered device-to-device from the dataset;
      * the means-only inference ``forward`` (block level: the same HIP kernels);
      * ``dv_recon_row_stats`` / ``dv_col_moments`` and their float64 combination, the Gaussian log-likelihood mean;
      * accuracy / AUROC / AUPR by sort + scan on the device (``metrics.*_dev``: no compaction, no host decisions);
    every scalar lands in one float64 vector.  ``run()`` = set the annealing coefficient, replay, ONE copy to the host.
    The group index lists (rows with a second profile / a label) are data of the DATASET, taken once; the arrays
    themselves are read by the replay, so a dataset edited in place is evaluated as edited."""

    @staticmethod
    def get(model, ds):
        x1 = getattr(ds, 'x1', None)
        if not (torch.is_tensor(x1) and x1.is_cuda) or getattr(model, 'use_s', False) \
                or getattr(model, 'type_y', 'discrete') != 'discrete' or getattr(model, 'type_rec', 'diag_gaussian') != 'diag_gaussian':
            return None
        if next(model.parameters()).device != x1.device:
            return None
        need = ('x1',) + (('x2', 'has_x2') if model.kind != 'vfae' else ()) + (('y', 'has_y') if model.kind != 'pvae' else ())
        parts = [getattr(ds, k, None) for k in need]
        if not all(torch.is_tensor(t) and t.device == x1.device for t in parts):
            return None          # (a host-resident label / flag array would be a pageable copy under capture)
        cache = model.__dict__.setdefault('_eval_graphs', {})
        # the captured launches point into the engine's arena, plan and layer chains: ``.cpu().cuda()`` / ``.to()`` retire
        # the engine (DGMMixin._apply) and a graph of the old one would read freed parameters -- the engine's identity is
        # part of the signature (and ``_apply`` drops the cache)
        eng = model.engine()
        # data parallelism: the ROWS of a whole-set evaluation are sharded over the ranks (every rank holds the dataset; its
        # heavy passes run on rows [n r / W, n (r + 1) / W) only), the partials meet in ONE all-reduce -- see ``_sequence_partial``
        dp = model._dp if (getattr(model, '_dp', None) is not None and getattr(model, 'shard_evaluation', True)
                           and int(x1.shape[0]) >= 2 * model._dp[1]) else None
        sig = tuple(int(t.data_ptr()) if torch.is_tensor(t) else 0
                    for t in (getattr(ds, k, None) for k in ('x1', 'x2', 'y', 'has_x2', 'has_y'))) + \
            (int(x1.shape[0]), id(eng), int(eng.arena.param.data_ptr()), dp)
        ev = cache.get(id(ds))
        if ev is None or ev.sig != sig:
            if len(cache) > 8:
                cache.clear()
            try:
                ev = cache[id(ds)] = _EvalGraph(model, ds, sig, dp=dp)
            except Exception as e:      # the reference evaluates on regardless of a failing loss pass (src/DrVAE.py:647-654):
                import warnings         # so does the step-by-step path, which this dataset now takes
                warnings.warn('drvae_amd: the captured whole-set evaluation could not be built (%s: %s); evaluating step '
                              'by step' % (type(e).__name__, e))
                ev = cache[id(ds)] = _EvalGraph.__new__(_EvalGraph)
                ev.sig, ev.graph = sig, None
        return ev if ev.graph is not None else None

    def __init__(self, model, ds, sig, dp=None):
        self.model, self.sig = model, sig
        self.graph = None
        kind = model.kind
        dev = ds.x1.device
        self.dp, self.full = dp, None
        if dp is not None:
            # this rank's rows as VIEWS of the dataset's tensors; the normalisers of the loss pass are the whole set's counts,
            # the Philox draws are keyed by the row's position in the whole set (``row0``): the shards' terms add up to the
            # one-rank evaluation's
            import types
            from . import dist as D
            rank, world = dp
            n_all = int(ds.x1.shape[0])
            self.lo, self.hi = n_all * rank // world, n_all * (rank + 1) // world
            self.full = ds
            hx = getattr(ds, 'has_x2', None) if kind != 'vfae' else None
            hy = getattr(ds, 'has_y', None) if kind != 'pvae' else None
            zer = np.zeros(n_all, np.int64)
            eng0 = model.engine()
            self.counts = D.global_counts(hx.cpu().numpy() if hx is not None else zer, hy.cpu().numpy() if hy is not None else zer,
                                          eng0.cfg.kind, eng0.cfg.semi_supervised, local=True)
            ds = types.SimpleNamespace(**{k: (getattr(ds, k)[self.lo:self.hi] if torch.is_tensor(getattr(ds, k, None)) else None)
                                          for k in ('x1', 'x2', 's', 'y', 'has_x2', 'has_y')})
        self.ds = ds
        g = lambda k: getattr(ds, k, None)
        # rows of the groups (once per dataset: a host sync each)
        self.x2idx = torch.nonzero(g('has_x2').reshape(-1).to(dev)).reshape(-1) if kind != 'vfae' else None
        self.yidx = torch.nonzero(g('has_y').reshape(-1).to(dev)).reshape(-1) if kind != 'pvae' else None
        self.x2idx32 = self.x2idx.to(torch.int32) if self.x2idx is not None else None
        self.yidx32 = self.yidx.to(torch.int32) if self.yidx is not None else None
        # the loss plan of the whole set: built by the ordinary path (host index lists), then reused
        eng = model.engine()
        keep, keep_training, was_training = eng.plan, eng.training, model.training
        keep_counts, keep_row0 = model.__dict__.get('_global_counts'), eng.row0
        try:
            model.eval()
            kw = dict(x1=g('x1'), s=g('s'))
            if kind != 'vfae':
                kw.update(x2=g('x2'), has_x2=g('has_x2'))
            if kind != 'pvae':
                kw.update(y=g('y'), has_y=g('has_y'))
            self.kw = kw
            if dp is not None:
                model._global_counts, model._row0_override = self.counts, self.lo
            model.run_on_batch(train_mode=False, **kw)          # (also the warm-up of every kernel of the sequence)
            self.plan = eng.plan
            n_in = int(ds.x1.shape[0])
            rows = np.asarray(self.plan.rows)
            self.sel = None if (len(rows) == n_in and (rows == np.arange(n_in)).all()) else torch.as_tensor(rows, device=dev)
            if dp is not None:
                self._shard_layout()
            self._sequence()                                    # warm-up of the rest (allocations, code objects)
            torch.cuda.synchronize()
            gph = torch.cuda.CUDAGraph()
            eng.join_side()
            with torch.cuda.graph(gph):
                self.names, self.vec, self.loss_keys = self._sequence()
            if dp is not None:
                self._finalize_full()                           # (warm-up)
                torch.cuda.synchronize()
                self.graph_final = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_final):
                    self._finalize_full()
            self.graph = gph
        finally:
            eng.plan, eng.training = keep, keep_training
            model.train(was_training)
            if dp is not None:
                model.__dict__.pop('_row0_override', None)
                if keep_counts is None:
                    model.__dict__.pop('_global_counts', None)
                else:
                    model._global_counts = keep_counts
                eng.row0 = keep_row0

    def _sequence(self):
        """the launch sequence (eager for the warm-up, then under capture); -> (names, float64 vector, loss keys)"""
        from . import kernels as K
        m, ds, kind = self.model, self.ds, self.model.kind
        eng, p = m.engine(), self.plan
        eng.plan = p
        dev = ds.x1.device
        # --- evaluation-mode losses on the whole set: the fused forward on this plan, inputs from the dataset
        p.feed_active = False
        x1 = ds.x1.to(torch.float32)
        p.XSRC[:p.B].copy_(x1.index_select(0, self.sel) if self.sel is not None else x1)
        if eng.cfg.has_pert:
            x2 = ds.x2.to(torch.float32)
            p.XSRC[p.B:].copy_(x2.index_select(0, self.sel) if self.sel is not None else x2)
        eng.training = False
        eng.draw_noise()
        eng.forward()
        losses = m._loss_tensors(eng)
        # every scalar of the evaluation lands in ONE float64 vector: [losses | y metrics | x1 metrics | x2 metrics]
        names = ['loss_' + k for k in sorted(E.LOSS_IDX, key=E.LOSS_IDX.get)]      # (all seven; ``run`` picks the model's)
        n_loss = len(names)
        if kind != 'pvae':
            names += ['y_auroc', 'y_aupr', 'y_acc']
        names += ['x1_' + k for k in _NAN4]
        has_x2 = kind != 'vfae' and (len(self.x2idx) > 0 if self.dp is None else self.full_n_x2 > 0)
        if has_x2:
            names += ['x2_' + k for k in _NAN4]
        if not hasattr(self, 'vec'):
            self.vec = torch.zeros(len(names), dtype=torch.float64, device=dev)
        vec = self.vec
        vec[:n_loss].copy_(eng.arena.loss[:n_loss])
        o = n_loss
        # --- means-only inference + metrics (the tail: dv_rank_metrics / dv_recon_finalize, no sort, no host decisions)
        res = self._infer()
        if self.dp is not None:
            self._sequence_partial(res, vec[:n_loss])
            return names, vec, list(losses)
        if kind != 'pvae':
            self._y_metrics(res, vec[o:o + 3])
            o += 3
        self._recon(ds.x1, res['px1'][0], res['px1'][1], None, vec[o:o + 4], 'x1', res['px_bias'])
        o += 4
        if has_x2:
            self._recon(ds.x2, res['px2'][0], res['px2'][1], self.x2idx32, vec[o:o + 4], 'x2', res['px_bias'])
        return names, vec, list(losses)

    # ------------------------------------------------------------ rows sharded over ranks (data parallelism, round 6)
    def _shard_layout(self):
        """ONE flat float64 exchange buffer for the whole evaluation, laid out for the WHOLE set: [loss scalars | proba | pred |
        x1: row statistics, log-likelihood rows, column-moment blocks of every rank | x2: the same].  Every rank writes its
        rows' slots (zeros elsewhere), one sum all-reduce leaves the complete arrays on every rank (float32 / int32 values are
        exact in float64, and x + 0 = x), and the finalising launches -- the same kernels as without sharding -- run on them."""
        from . import kernels as K
        m, full, kind = self.model, self.full, self.model.kind
        dev = full.x1.device
        rank, world = self.dp
        n, X, Y = int(full.x1.shape[0]), int(full.x1.shape[1]), (m.dim_y if kind != 'pvae' else 0)
        self.n_all = n
        bounds = [(n * r // world, n * (r + 1) // world) for r in range(world)]
        self.full_x2idx32 = self.full_yidx32 = None
        self.full_n_x2 = 0
        blk2 = 1
        if kind != 'vfae':
            hx = full.has_x2.reshape(-1).to(dev) != 0
            self.full_x2idx32 = torch.nonzero(hx).reshape(-1).to(torch.int32)
            self.full_n_x2 = int(self.full_x2idx32.numel())
            per = [int(hx[a:b].sum()) for a, b in bounds]
            blk2 = max(K.col_moment_blocks(max(c, 1)) for c in per)
        if kind != 'pvae':
            self.full_yidx32 = torch.nonzero(full.has_y.reshape(-1).to(dev)).reshape(-1).to(torch.int32)
        blk1 = max(K.col_moment_blocks(b - a) for a, b in bounds)
        o, self.off = 0, {}
        for name, size in (('loss', 8), ('proba', n * Y), ('pred', n if Y else 0), ('rows1', n * 6), ('ll1', n),
                           ('part1', world * blk1 * 3 * X), ('rows2', n * 6 if self.full_n_x2 else 0),
                           ('ll2', n if self.full_n_x2 else 0), ('part2', world * blk2 * 3 * X if self.full_n_x2 else 0)):
            self.off[name] = (o, size)
            o += size
        self.blk = {'x1': blk1, 'x2': blk2}
        self.pack = torch.zeros(o, dtype=torch.float64, device=dev)
        f32 = lambda *shape: torch.zeros(*shape, device=dev)
        self.g_rows = {'x1': f32(n, 6), 'x2': f32(n, 6)}
        self.g_ll = {'x1': f32(n), 'x2': f32(n)}
        if Y:
            self.g_proba, self.g_pred32 = f32(n, Y), torch.zeros(n, dtype=torch.int32, device=dev)
            self.g_y32 = torch.zeros(n, dtype=torch.int32, device=dev)

    def _slot(self, name):
        o, size = self.off[name]
        return self.pack[o:o + size]

    def _sequence_partial(self, res, loss_vec):
        """this rank's share of the evaluation's partials into the exchange buffer (captured; no finalising launch)"""
        from . import kernels as K
        from ._lib import GAUSS_SIGMA
        m, ds, kind = self.model, self.ds, self.model.kind
        rank, world = self.dp
        lo, hi, X = self.lo, self.hi, int(ds.x1.shape[1])
        self.pack.zero_()
        self._slot('loss')[:loss_vec.numel()].copy_(loss_vec)
        if kind != 'pvae':
            Y = m.dim_y
            self._slot('proba').view(self.n_all, Y)[lo:hi].copy_(res['proba'])
            self._slot('pred')[lo:hi].copy_(res['pred'].reshape(-1))
        bias = res['px_bias']
        for tag, x, key, sel in (('x1', ds.x1, 'px1', None), ('x2', getattr(ds, 'x2', None), 'px2', self.x2idx32)):
            if tag == 'x2' and (kind == 'vfae' or not self.full_n_x2):
                continue
            x = x.to(torch.float32)
            M = int(x.shape[0])
            n_sel = int(sel.numel()) if sel is not None else M
            buf = self.__dict__.setdefault('_recon_bufs', {})
            if tag not in buf:
                buf[tag] = (torch.empty(M, 6, device=x.device),
                            torch.zeros(self.blk[tag], 3, X, dtype=torch.float64, device=x.device), torch.empty(M, device=x.device))
            rows, part, ll = buf[tag]
            x_rec, x_std = res[key][0], res[key][1]
            if X <= K.RECON_ROWS_MAX_X:
                K.recon_rows(rows, ll, x, x_rec, x_std, bias=bias[:2] if bias is not None else None,
                             sd_shift=bias[2] if bias is not None else 0.0)
            else:
                K.recon_row_stats(rows, x, x_rec)
                K.nll_rows_fwd(ll, x, x_rec, x_std, mode=GAUSS_SIGMA)
            t = '1' if tag == 'x1' else '2'
            self._slot('rows' + t).view(self.n_all, 6)[lo:hi].copy_(rows)
            self._slot('ll' + t)[lo:hi].copy_(ll)
            if n_sel > 0:
                nb = K.col_moment_blocks(n_sel)
                K.col_moments(None, x, x_rec, sel=sel, part=part[:nb], r_bias=bias[0] if bias is not None else None)
                self._slot('part' + t).view(world, self.blk[tag], 3, X)[rank, :nb].copy_(part[:nb])

    def _finalize_full(self):
        """behind the all-reduce: the complete arrays out of the exchange buffer, then the SAME finalising launches as the
        unsharded evaluation (``dv_rank_metrics``, ``dv_recon_finalize``) over the whole set -> ``vec``"""
        from . import kernels as K
        m, kind, full = self.model, self.model.kind, self.full
        vec, n, X = self.vec, self.n_all, int(full.x1.shape[1])
        n_loss = len(E.LOSS_IDX)
        vec[:n_loss].copy_(self._slot('loss')[:n_loss])
        o = n_loss
        if kind != 'pvae':
            Y = m.dim_y
            self.g_proba.copy_(self._slot('proba').view(n, Y))
            self.g_pred32.copy_(self._slot('pred'))
            self.g_y32.copy_(full.y.reshape(-1))      # (read by the replay: a dataset edited in place is evaluated as edited)
            self._rank_metrics(self.g_proba, self.g_y32, self.g_pred32, self.full_yidx32, vec[o:o + 3], n)
            o += 3
        for tag, t, sel in (('x1', '1', None), ('x2', '2', self.full_x2idx32)):
            if tag == 'x2' and (kind == 'vfae' or not self.full_n_x2):
                continue
            self.g_rows[tag].copy_(self._slot('rows' + t).view(n, 6))
            self.g_ll[tag].copy_(self._slot('ll' + t))
            part = self._slot('part' + t).view(-1, 3, X)
            K.recon_finalize(vec[o:o + 4], self.g_rows[tag], part, X, sel=sel, n=int(sel.numel()) if sel is not None else n,
                             ll=self.g_ll[tag])
            o += 4

    def _rank_metrics(self, proba, y32, pred32, yidx32, out3, n_rows):
        """[auroc, aupr, acc] of the labeled rows ``yidx32`` from the pair-counting kernel (sort + scan beyond its range)"""
        from . import kernels as K
        m = self.model
        n_lab, Y = int(yidx32.numel()), m.dim_y
        if n_lab > K.RANK_MAX_ROWS or n_lab == 0:
            idx = yidx32.long()
            v = MET.eval_y_prediction_dev(pred32.index_select(0, idx), proba.index_select(0, idx), y32.index_select(0, idx).long(), Y)
            out3.copy_(torch.stack([v['auroc'].reshape(()), v['aupr'].reshape(()), v['acc'].reshape(())]))
            return
        if not hasattr(self, 'rank_counts'):
            n_cls = 1 if Y == 2 else Y
            self.rank_counts = torch.zeros(n_cls, n_lab, 4, dtype=torch.int32, device=proba.device)
            self.rank_out = torch.zeros(2 * n_cls + 1, dtype=torch.float64, device=proba.device)
        if Y == 2:
            K.rank_metrics(out3, self.rank_counts, proba, y32, pred32=pred32, sel=yidx32, c0=1, n_cls=1, binary=True)
        else:
            K.rank_metrics(self.rank_out, self.rank_counts, proba, y32, pred32=pred32, sel=yidx32, c0=0, n_cls=Y, binary=False)
            out3[0:1].copy_(self.rank_out[0:2 * Y:2].mean().reshape(1))
            out3[1:2].copy_(self.rank_out[1:2 * Y:2].mean().reshape(1))
            out3[2:3].copy_(self.rank_out[2 * Y:2 * Y + 1])

    def _y_metrics(self, res, out3):
        """accuracy / ROC-AUC / average precision of the labeled rows (src/DGMMixin.py:158-190) -> out3 = [auroc, aupr, acc]"""
        from . import kernels as K
        m, ds = self.model, self.ds
        dev = ds.x1.device
        n_lab, Y = int(self.yidx.numel()), m.dim_y
        if n_lab > K.RANK_MAX_ROWS or n_lab == 0:      # (beyond the pair-counting kernel's range: the sort + scan formulation)
            y = ds.y.to(dev)
            ylab = y.reshape(-1).index_select(0, self.yidx)
            v = MET.eval_y_prediction_dev(res['pred'].index_select(0, self.yidx), res['proba'].index_select(0, self.yidx), ylab, Y)
            out3.copy_(torch.stack([v['auroc'].reshape(()), v['aupr'].reshape(()), v['acc'].reshape(())]))
            return
        if not hasattr(self, 'y32'):
            n_cls = 1 if Y == 2 else Y
            self.y32 = torch.zeros(int(ds.x1.shape[0]), dtype=torch.int32, device=dev)
            self.pred32 = torch.zeros_like(self.y32)
            self.rank_counts = torch.zeros(n_cls, n_lab, 4, dtype=torch.int32, device=dev)
            self.rank_out = torch.zeros(2 * n_cls + 1, dtype=torch.float64, device=dev)
        self.y32.copy_(ds.y.reshape(-1))
        self.pred32.copy_(res['pred'].reshape(-1))
        proba = res['proba']
        if Y == 2:
            K.rank_metrics(out3, self.rank_counts, proba, self.y32, pred32=self.pred32, sel=self.yidx32, c0=1, n_cls=1, binary=True)
        else:       # macro average over the one-vs-rest problems (nan as soon as one class's value is: the mean propagates it)
            K.rank_metrics(self.rank_out, self.rank_counts, proba, self.y32, pred32=self.pred32, sel=self.yidx32, c0=0,
                           n_cls=Y, binary=False)
            out3[0:1].copy_(self.rank_out[0:2 * Y:2].mean().reshape(1))
            out3[1:2].copy_(self.rank_out[1:2 * Y:2].mean().reshape(1))
            out3[2:3].copy_(self.rank_out[2 * Y:2 * Y + 1])

    def _infer(self):
        """``forward`` (means-only inference, src/DrVAE.py:253-311) with its two heavy blocks on the fused step's layer
        chains: the encoder's heads as ONE (mu | logvar) product, and the decoder ONCE over the stacked rows [z1; z2]
        with its (mu | std) heads as one product -- 3 launches of 2n rows instead of 6 of n (the block modules compute
        every head with a launch of its own).  The small blocks in between are the model's own modules."""
        from .chain import _Chain
        from . import kernels as K, tuning as T
        m, ds, kind = self.model, self.ds, self.model.kind
        eng = m.engine()
        n, Z, X = int(ds.x1.shape[0]), eng.cfg.dim_z1, eng.cfg.dim_x
        dev = ds.x1.device
        if not hasattr(self, 'c_dec'):
            if self.sel is not None:
                self.c_enc = _Chain(eng.L_enc, n, dev)
            self.c_dec = _Chain(eng.L_decx, n * (2 if kind != 'vfae' else 1), dev)
            self.zd = torch.zeros(n * (2 if kind != 'vfae' else 1), (Z + 3) // 4 * 4, device=dev)[:, :Z]
        if self.sel is None:
            # q(z1|x1) of every row has just been computed by the loss pass (``_sequence`` runs it first): evaluation mode adds
            # no input noise, its encoder input rows [0, n) ARE x1 in dataset order, same layer chain, same kernels -- the
            # first n rows of its heads' buffer are this product (8192 rows: 15 GFLOP not computed twice)
            Q = self.plan.c_enc.out[-1][:n]
        else:
            x1 = ds.x1.to(torch.float32)
            if X % 4:
                # rows padded to 16 B (zero pads): the first layer's product then runs on the LDS-DMA kernels (``_Chain``)
                if not hasattr(self, 'x1p'):
                    self.x1p = torch.zeros(n, (X + 3) // 4 * 4, device=dev)[:, :X]
                self.x1p.copy_(x1)
                x1 = self.x1p
            Q = self.c_enc.forward([x1])
        z1 = Q[:, :Z]
        res = OrderedDict(z1=z1, qz1=(z1, Q[:, Z:2 * Z]))
        if kind != 'vfae':
            # the perturbation function's MEAN only (its log-variance feeds no metric): z2 = z1 + z1 W_mu^T + b
            # (src/blocks.py:357), one launch, straight into the second half of the decoder's stacked input
            lz = eng.L_z2F[0]
            z2 = self.zd[n:]
            K.linear_fwd(z2, z1, lz.W[:Z], lz.b[:Z], resid=z1, resid_cols=Z, overread=True)
            res.update(z2=z2)
        if kind != 'pvae':
            if kind == 'drvae':
                clf_in = [z1, z2 - z1] if m.clf_z1z2 else [z2]
            else:
                clf_in = [z1]
            res.update(**m._pred_proba(m.encoder_y(clf_in)))
        self.zd[:n].copy_(z1)
        # the decoder's heads as a PLAIN product where the one-pass reconstruction statistics finish them on their way
        # (bias, softplus + shift: ``dv_recon_rows`` / ``dv_col_moments``): 16384 x 1956 x 600 at the raw product's 126 instead of 104 TF/s
        lh = eng.L_decx[-1]
        raw = bool(X <= K.RECON_ROWS_MAX_X and self.c_dec.raw_last_ok() and lh.act1 == 'softplus' and T.get('raw_heads'))
        PX = self.c_dec.forward([self.zd], raw_last=raw)
        res['px_bias'] = (lh.b[:X], lh.b[X:2 * X], lh.shift1) if raw else None
        res['px1'] = (PX[:n, :X], PX[:n, X:2 * X])
        if kind != 'vfae':
            res['px2'] = (PX[n:, :X], PX[n:, X:2 * X])
        return res

    def _recon(self, x, x_rec, x_std, sel, out4, tag, bias=None):
        """``eval_x_reconstruction`` (src/DGMMixin.py:128-156) over the rows ``sel`` (all when None): row statistics, column
        moments and log-likelihood rows from the HIP kernels, combined in float64 by ``dv_recon_finalize`` into ``out4`` =
        [rmse, r2, pearr, ll].  The statistics are taken over ALL rows and selected afterwards (no gathered copies of
        the x2 rows and their reconstructions).  Rows of up to 1024 genes: row statistics and log-likelihood rows in ONE pass
        (``dv_recon_rows``); ``bias`` = (bias_mu, bias_sd, shift): the heads are raw products, finished by the passes."""
        from . import kernels as K
        from ._lib import GAUSS_SIGMA
        x = x.to(torch.float32)
        M, X = x.shape
        n = int(sel.numel()) if sel is not None else M
        one_pass = X <= K.RECON_ROWS_MAX_X
        assert bias is None or one_pass
        buf = self.__dict__.setdefault('_recon_bufs', {})
        if tag not in buf:
            buf[tag] = (torch.empty(M, 6, device=x.device), torch.empty(K.col_moment_blocks(n), 3, X, dtype=torch.float64, device=x.device),
                        torch.empty(M, device=x.device))
        rows, part, ll = buf[tag]
        if one_pass:
            K.recon_rows(rows, ll, x, x_rec, x_std, bias=bias[:2] if bias is not None else None,
                         sd_shift=bias[2] if bias is not None else 0.0)
        else:
            K.recon_row_stats(rows, x, x_rec)
            K.nll_rows_fwd(ll, x, x_rec, x_std, mode=GAUSS_SIGMA)
        K.col_moments(None, x, x_rec, sel=sel, part=part, r_bias=bias[0] if bias is not None else None)
        K.recon_finalize(out4, rows, part, X, sel=sel, n=n, ll=ll)

    def run(self):
        m, kind = self.model, self.model.kind
        eng = m.engine()
        eng.join_side()
        eng.iters = m.finished_training_iters
        self.plan.set_beta(eng.beta_pert())                 # (the annealing coefficient is data of the plan, not of the graph)
        self.graph.replay()
        eng._noise_stale = True      # (the replay drew from the Philox counter: a train step drawn ahead re-draws, as after any eager draw)
        if self.dp is not None:      # rows sharded over the ranks: ONE sum all-reduce of the partials, then the finalising launches
            from . import dist as D
            D.allreduce_sum(self.pack)
            self.graph_final.replay()
        v = dict(zip(self.names, self.vec.cpu().tolist()))  # THE host sync of the evaluation
        perf = OrderedDict()
        perf['losses'] = OrderedDict((k, torch.tensor(v['loss_' + k])) for k in self.loss_keys)
        parts = []
        if kind != 'pvae':
            for k in ('acc', 'auroc', 'aupr'):
                perf['y_' + k] = v['y_' + k]
            parts.append('Y: Accuracy: {:.3f}% AUROC: {:.3f} AUPR: {:.3f}'.format(perf['y_acc'] * 100., perf['y_auroc'], perf['y_aupr']))
        for k in _NAN4:
            perf['x1_' + k] = v['x1_' + k]
        parts.append('X1: ' + _REC.format(perf['x1_rmse'], perf['x1_r2'], perf['x1_pearr']))
        if kind != 'vfae':
            if 'x2_rmse' in v:
                for k in _NAN4:
                    perf['x2_' + k] = v['x2_' + k]
                parts.append('X2: ' + _REC.format(perf['x2_rmse'], perf['x2_r2'], perf['x2_pearr']))
            else:
                for k in _NAN4:
                    perf['x2_' + k] = np.nan
                parts.append('X2: no x2 data')
        perf['model_class'] = m.__class__.__name__
        return perf, '\t '.join(parts)
