"""Train-step mixin shared by the three models -- counterpart of the hot-path part of
reference ``src/DGMMixin.py`` (``_create_optimizer``, ``_use_free_bits``,
``_compute_anneal_coef``, ``run_on_batch``, ``save_to_file``/``load_params_from_file``).

Differences by design: the optimizer is not a ``torch.optim`` object but the fused Adam
kernel over the flat parameter arena, and ``run_on_batch(train_mode=True)`` is the fused
forward+backward+Adam launch sequence of ``drvae_amd.engine`` (optionally replayed from a
hipGraph) instead of autograd.  ``fit``, early stopping and the prediction metrics live in
``drvae_amd/fit.py`` / ``drvae_amd/metrics.py``.
"""
from collections import OrderedDict

import torch

from . import engine as E
from .arena import ParamArena


class DeepGenerativeModelMixin:
    def w2log(self, *args):
        """print, and append to logs/<log_txt> when set (src/DGMMixin.py:20-29)."""
        if self.dp_rank != 0:          # data parallelism: rank 0 speaks for the job
            return
        if getattr(self, 'verbose_log', False):
            print(*args)
        if getattr(self, 'log_txt', None) is not None:
            import os
            os.makedirs('logs', exist_ok=True)
            with open(os.path.join('logs', self.log_txt), 'a') as f:
                f.write(' '.join(str(e) for e in args) + '\n')

    # ----------------------------------------------------------------- optimiser
    def _create_optimizer(self):
        """Adam with coupled L2 ``weight_decay`` on every parameter (src/DGMMixin.py:31-40),
        or Adamax (src/DGMMixin.py:37-38), as one fused kernel over the arena."""
        if self.optim_alg not in ('adam', 'adamax'):
            raise ValueError('Selected unknown optimizer: ' + str(self.optim_alg))
        self.optimizer = None          # kept for attribute parity; the state lives in the arena
        self._engine = None
        self.__dict__.pop('_eval_graphs', None)

    def _step_config(self):
        raise NotImplementedError

    # ------------------------------------------------------------ data parallelism
    _dp = None            # (rank, world) once ``enable_data_parallel`` has been called
    _allreduce = None     # the gradient exchange of a train step: callable(flat exchange buffer) | None

    def enable_data_parallel(self, rank=None, world=None, backend=None, broadcast=True):
        """Train this model data-parallel, one process per GPU (the reference has no distributed code).
        ``rank`` / ``world`` default to the environment of the launcher (RANK / WORLD_SIZE / MASTER_*: ``torchrun`` or
        ``python -m torch.distributed.run``); the process group is created if there is none (``backend``: 'nccl' = RCCL
        on GPUs, 'gloo' elsewhere).  From here on

        * ``run_on_batch(train_mode=True, ...)`` takes THIS RANK's rows of the global minibatch: the loss normalisers
          (N_total, N_pairs, N_labeled, src/DrVAE.py:611-616) are the global batch's (one tiny host all-reduce of the
          three counts), the flat buffer [loss scalars | gradients] is summed over the ranks by ONE all-reduce and every
          rank applies the identical fused Adam step -- the job trains like one process on the concatenated batch;
        * ``fit(DeviceBatcher(..., batch_size=rows per rank), ...)`` binds the batcher with ``dp=(rank, world)``: every
          rank draws the same global index table from the shared seed and runs its columns; the global counts of every
          batch are table data (no collective but the gradient exchange); captured split graphs + the exchange per step;
        * ``evaluate_performance_on_dataset`` of an HBM-resident dataset is sharded BY ROWS (round 6): every rank runs the loss
          pass and the inference on its n / world rows (global normalisers, Philox draws keyed by the row's position in
          the whole set), the partials -- loss scalars, per-row statistics, float64 column moments, class probabilities --
          meet in ONE all-reduce, and the same finalising launches as in a one-rank evaluation leave identical metrics on
          every rank, so early stopping decides the same everywhere (``model.shard_evaluation = False``: every rank
          evaluates the whole set); ``run_on_batch(train_mode=False)`` evaluates what it is given; only rank 0 writes
          snapshots and log lines.

        ``broadcast``: parameters, Adam moments and the step / Philox counters of rank 0 go to every rank first.
        Returns (rank, world).  A one-rank world (or none) leaves the model single-process unless DRVAE_FORCE_DP=1."""
        from . import dist as D
        if rank is None or world is None:
            rank, world, _ = D.init_from_env(backend)
        rank, world = int(rank), int(world)
        assert 0 <= rank < max(world, 1)
        if getattr(self, 'use_MMD', False) and world > 1:
            raise NotImplementedError('use_MMD: the model-level MMD penalty compares every row of a nuisance class with '
                                      'every other row of the batch: it cannot be sharded over ranks')
        active = world > 1 or D.force_dp()
        self._dp = (rank, world) if active else None
        self._allreduce = D.allreduce_sum if active else None
        if active and broadcast:
            eng = self.engine()
            eng.join_side()
            for t in (eng.arena.param, eng.arena.exp_avg, eng.arena.exp_avg_sq, eng.step_dev, eng.rng_ctr):
                D.broadcast(t)
            eng.sync_side_counters()
            self.finished_training_iters = int(D.broadcast_int(self.finished_training_iters))
        return rank, world

    def disable_data_parallel(self):
        self._dp = self._allreduce = None
        if getattr(self, '_engine', None) is not None:
            self._engine.row0 = 0

    @property
    def dp_rank(self):
        return self._dp[0] if self._dp else 0

    def engine(self):
        """Build (once) the parameter arena + fused step engine on the model's device."""
        if self._engine is None:
            dev = next(self.parameters()).device
            cfg = self._step_config()
            shapes = OrderedDict((k, tuple(v.shape)) for k, v in self.named_parameters())
            want = E.param_shapes(cfg)
            if list(shapes.items()) != list(want.items()):
                raise RuntimeError('model parameters do not match the fused-step layout: %s'
                                   % (set(shapes.items()) ^ set(want.items())))
            self._arena = ParamArena(shapes, dev, frozen=E.frozen_params(cfg)).adopt(self)
            self._engine = E.FusedStep(cfg, self._arena, seed=self.random_seed)
            self._restore_optimizer_state()
        return self._engine

    # ---- the nn.Parameters ALIAS the arena; anything that swaps ``prm.data`` (``.to()``, ``.cpu()``,
    # ``load_state_dict(assign=True)``) would leave the fused step training a buffer nobody reads
    def _arena_aliased(self):
        a = self._arena
        return all(prm.dtype == torch.float32 and prm.data_ptr() == a.p(name).data_ptr()
                   for name, prm in self.named_parameters())

    def _assert_arena_aliased(self):
        if getattr(self, '_engine', None) is not None and not self._arena_aliased():
            raise RuntimeError('drvae_amd: model parameters no longer alias the parameter arena of the fused step')

    def _stash_optimizer_state(self):
        eng, a = self._engine, self._arena
        self._opt_stash = dict(exp_avg=a.exp_avg.detach().cpu().clone(), exp_avg_sq=a.exp_avg_sq.detach().cpu().clone(),
                               step=eng.step_dev.cpu().clone(), rng=eng.rng_ctr.cpu().clone(), iters=eng.iters,
                               offsets=dict(a.offsets))

    def _restore_optimizer_state(self):
        st = getattr(self, '_opt_stash', None)
        if st is None:
            return
        eng, a = self._engine, self._arena
        assert st['offsets'] == a.offsets
        a.exp_avg.copy_(st['exp_avg'])
        a.exp_avg_sq.copy_(st['exp_avg_sq'])
        eng.step_dev.copy_(st['step'])
        eng.rng_ctr.copy_(st['rng'])
        eng.sync_side_counters()
        eng.iters = st['iters']
        self._opt_stash = None

    def _apply(self, fn, recurse=True):
        """``.to()/.cuda()/.cpu()/.float()``: when the parameters were moved off the arena, carry the
        optimiser state (Adam moments, step and Philox counters) over and rebuild the arena + engine on
        the new device at the next use.  A dtype other than fp32 is refused: the hot path is fp32."""
        if getattr(self, '_engine', None) is not None:
            probe = next(self.parameters())
            if fn(torch.empty(0, dtype=probe.dtype, device=probe.device)).dtype != torch.float32:
                raise TypeError('drvae_amd: the fused train step is fp32; cast a copy of the model instead')
        out = super()._apply(fn, recurse)
        if getattr(self, '_engine', None) is not None and not self._arena_aliased():
            self._stash_optimizer_state()
            self._engine = self._arena = None      # rebuilt (and the parameters re-adopted) by engine()
            self.__dict__.pop('_eval_graphs', None)    # (captured evaluations point into the retired arena / plans)
        return out

    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        if assign and getattr(self, '_engine', None) is not None and not self._arena_aliased():
            self._arena.adopt(self)                # values copied into the arena, parameters aliased again
        return out

    # ------------------------------------------------------- model-level MMD penalty
    def _get_mmd_criterion(self, z, sind):
        """Minus the MMD (``blocks.mmd_objective`` with ``self.kernel_MMD``) between the latent rows of every
        category of the nuisance variable s and the rows outside it, averaged over the categories; with two
        categories only the first pair (src/DGMMixin.py:42-66).  ``sind``: one 0/1 indicator vector per
        category.  A side without rows is replaced by one random N(0,1) row, like the reference.  As shipped
        the reference function cannot run (two missing imports); with those supplied its values are the
        golden vectors of tests/golden/mmd_criterion.npz.  The row selection is an index gather, so
        one-member categories work too (the reference's ``len()`` of a 0-d tensor raises there)."""
        from . import blocks as blk
        return blk.mmd_criterion(z, sind, self.kernel_MMD)

    # --------------------------------------------------------------- small helpers
    def _use_free_bits(self, KL_perx, override_default_kl_min=None):
        """max(KL_row, kl_min) on the per-row KL (src/DGMMixin.py:68-75)."""
        kl_min = self.kl_min if override_default_kl_min is None else override_default_kl_min
        return torch.clamp(KL_perx, min=float(kl_min))

    def _compute_anneal_coef(self, iter_num, iter_max=1000, iter_offset=0, func_type='linear'):
        if func_type != 'linear':
            raise ValueError('Unknown annealing function: ' + func_type)
        return E.anneal_coef(iter_num, iter_max, iter_offset)

    # -------------------------------------------------------------------- the step
    def _batch_to_engine(self, x1, x2=None, s=None, y=None, has_x2=None, has_y=None, dp=False):
        """``dp``: the rows are this rank's share of a global minibatch (train steps under ``enable_data_parallel``):
        global normalisers, Philox draws keyed by the row's global position; otherwise a batch of its own"""
        eng = self.engine()
        n = x1.shape[0]
        dev = eng.dev
        x1 = x1.to(dev, torch.float32).contiguous()
        x2 = x2.to(dev, torch.float32).contiguous() if x2 is not None else None
        zeros = torch.zeros(n, dtype=torch.int64)
        has_x2 = has_x2 if has_x2 is not None else zeros
        has_y = has_y if has_y is not None else zeros
        counts = getattr(self, '_global_counts', None)
        eng.row0 = int(getattr(self, '_row0_override', 0))      # (a row shard of a whole-set evaluation: ``fit._EvalGraph``)
        if dp and self._dp is not None:
            from . import dist as D
            eng.row0 = self._dp[0] * n          # (every rank feeds the same number of rows: ``shard_rows``)
            if counts is None:
                counts = D.global_counts(has_x2.cpu() if torch.is_tensor(has_x2) else has_x2,
                                         has_y.cpu() if torch.is_tensor(has_y) else has_y, eng.cfg.kind,
                                         eng.cfg.semi_supervised)
        eng.set_batch(x1, x2, y, has_x2, has_y, counts=counts, s=s if eng.cfg.use_s else None)
        return eng

    def _loss_tensors(self, eng):
        loss = eng.arena.loss
        keys = list(E.LOSS_IDX)
        if eng.cfg.kind == 'pvae':
            keys.remove('YL')
        if eng.cfg.kind == 'vfae':
            keys.remove('PERT')
        return OrderedDict((k, loss[E.LOSS_IDX[k]]) for k in keys)

    def loss_function(self, noise=None, **kwargs):
        """The 7 loss scalars (RECL, KLD, PERT, YL, MMD, ELBO, CMPL) of src/DrVAE.py:545-626 as
        0-d device tensors (views: valid until the next call).  Forward only."""
        eng = self._batch_to_engine(**kwargs)
        eng.training = self.training
        eng.add_noise = bool(getattr(self, 'add_noise', False))
        eng.iters = self.finished_training_iters
        if noise is not None:
            eng.set_noise(noise)
        else:
            eng.draw_noise()
        eng.forward()
        return self._loss_tensors(eng)

    def run_on_batch(self, train_mode=False, noise=None, **kwargs):
        """Train / evaluate on one minibatch (src/DGMMixin.py:91-126).  ``noise`` optionally
        injects the N(0,1) draws (parity tests); by default they come from on-device Philox."""
        if not train_mode:
            self.eval()
            return self.loss_function(noise=noise, **kwargs)
        self.train()
        eng = self._batch_to_engine(dp=True, **kwargs)
        eng.add_noise = bool(getattr(self, 'add_noise', False))
        eng.iters = self.finished_training_iters
        eng.train_step(noise, allreduce=self._allreduce)
        self.finished_training_iters = eng.iters
        return self._loss_tensors(eng)

    # ---------------------------------------------------------------- evaluation
    @torch.no_grad()
    def eval_x_reconstruction(self, x, x_rec, x_rec_logvar=None):
        """RMSE, variance-weighted R^2, mean per-row Pearson r and mean log-likelihood of a
        reconstruction (src/DGMMixin.py:128-156).  The O(rows x genes) reductions run in two HIP
        kernels; the final combination of the (rows x 6) / (3 x genes) partials is done in float64
        on the host, like the reference's numpy/scipy/sklearn code.  ``x_rec_logvar`` is what the
        reference passes as third argument of ``decoder_x.logp_perx``: the decoder's std."""
        from . import kernels as K
        dev = next(self.parameters()).device
        x = x.to(dev, torch.float32).contiguous()
        x_rec = x_rec.to(dev, torch.float32).contiguous()
        M, X = x.shape
        rows = torch.empty(M, 6, device=dev)
        cols = torch.empty(3, X, dtype=torch.float64, device=dev)
        K.recon_row_stats(rows, x, x_rec)
        K.col_moments(cols, x, x_rec)
        r, c = rows.double().cpu().numpy(), cols.cpu().numpy()
        import numpy as np
        out = dict()
        out['rmse'] = float(np.sqrt(r[:, 0].sum() / (M * X)))
        ss_tot = c[1] - c[0] ** 2 / M
        out['r2'] = float(1.0 - c[2].sum() / ss_tot.sum())
        with np.errstate(divide='ignore', invalid='ignore'):
            out['pearr'] = float((r[:, 5] / np.sqrt(r[:, 3] * r[:, 4])).mean())
        if x_rec_logvar is not None:
            out['ll'] = float(self.decoder_x.logp_perx(x, x_rec, x_rec_logvar.to(dev, torch.float32)).mean())
        else:
            out['ll'] = float('nan')
        return out

    # ---------------------------------------------------------------- checkpoints
    def save_to_file(self, filename):
        """``torch.save(state_dict)`` with the reference's key names (src/DGMMixin.py:192-197)."""
        torch.save(OrderedDict((k, v.detach().cpu().clone()) for k, v in self.state_dict().items()), filename)

    def load_params_from_file(self, filename):
        self.load_state_dict(torch.load(filename, map_location='cpu'))
