"""How the launch sequence of the fused train step is put on the GPU: hipGraph capture / replay, the
dual-graph schedule (main chain and side chain as two single-stream graphs ordered by device flags),
the CU partition of the two chains, and the data-parallel graph splits.  Mixed into ``FusedStep``."""
import ctypes
import gc
import os
import weakref

import torch

from . import kernels as K
from . import tuning as T


def _masked_stream(bits, device):
    """a HIP stream whose kernels only run on the CUs set in ``bits`` (hipExtStreamCreateWithCUMask),
    wrapped for torch; lives for the rest of the process"""
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(bits))(*bits)
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(bits), arr)
    if rc != 0:
        raise RuntimeError('hipExtStreamCreateWithCUMask failed: %d' % rc)
    import atexit

    def _destroy(handle=st.value):       # before the HIP runtime tears down (profilers crash otherwise)
        try:
            torch.cuda.synchronize()
            hip.hipStreamDestroy(ctypes.c_void_p(handle))
        except Exception:
            pass
    atexit.register(_destroy)
    return torch.cuda.ExternalStream(st.value, device=device)


SYNC_POLL = T.get('sync_poll')     # steps between two polls of the wait-error words

_PINNED_POOL = []             # pinned int32 buffers of retired engines (see ``_poll_sync``)
_PARTITION_STREAMS = {}      # device index -> {reserved CUs -> (main stream, side stream) | None}


class _Branch:
    """Fork/join of an independent launch chain onto a side HIP stream.  Inside a hipGraph
    capture the side stream joins the capture, so the chain becomes a parallel branch of the
    graph; on CPU tensors (unit tests with stand-in launchers) it degrades to inline execution."""

    def __init__(self, device, enabled=True):
        self.on = enabled and torch.device(device).type == 'cuda'
        self.side = torch.cuda.Stream(device=device) if self.on else None

    def fork(self):
        """mark the point of the current stream the side chain depends on; the chain itself may be
        recorded later (``with branch:``).  Measured on MI355X (tools/graph_fork_probe.py): the hipGraph
        executor runs a fork/join ~30 us faster when the MAIN continuation is recorded before the side
        chain, so callers fork, record the main work, and only then the side chain."""
        if self.on:
            self.side.wait_stream(torch.cuda.current_stream())
            self._forked = True

    def __enter__(self):
        if self.on:
            if not getattr(self, '_forked', False):
                self.side.wait_stream(torch.cuda.current_stream())
            self._forked = False
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self._ctx.__exit__(*exc)

    def join(self):
        if self.on:
            torch.cuda.current_stream().wait_stream(self.side)


# what ``capture()`` leaves on the engine: kept per plan by ``stash_capture`` / swapped back in by ``use_capture``
_CAPTURE_STATE = ('_graphs', '_side_graph', '_split_capture', '_split_kind', 'noise_ahead', '_graph_key', '_graph_feed',
                  '_graph_mmd_sig', '_captured_allreduce', '_graph_noise')


class StepSchedule:
    """scheduling half of ``FusedStep`` (see the module docstring); relies on its forward / backward /
    optimizer_step / draw_noise and on its plan, arena and counters"""

    # ------------------------------------------------ several captured plans side by side
    def stash_capture(self):
        """remember the graphs just captured for the current plan (a feed that switches between a few plans from step
        to step -- ``DeviceBatcher(pair_bucket=...)`` -- captures each once and swaps them with ``use_capture``)"""
        if not hasattr(self, '_captures'):
            self._captures = {}
        assert self._graph_key == self.plan.key, 'stash_capture right after capture(): another plan is current'
        self._captures[self.plan.key] = {k: getattr(self, k, None) for k in _CAPTURE_STATE}

    def use_capture(self, key):
        """make the plan ``key`` and its captured graphs current; False when it has not been captured yet"""
        cap = getattr(self, '_captures', {}).get(key)
        if cap is None or cap['_graph_feed'] is not self._plans[key].live_feed or \
                cap['_graph_noise'] not in (None, self.add_noise):
            return False
        if getattr(self, '_graph_key', None) != key:
            for k, v in cap.items():
                setattr(self, k, v)
            self.plan = self._plans[key]
            # the side chain drew ahead into the OTHER plan's noise buffer: the replay draws for this one first (in front
            # of the main graph; drawn on the side stream with an event edge instead, the step is 2 % slower)
            self._noise_stale = True
        return True

    def _mode(self):
        if not self.fuse_bwd:
            return 0
        if self._rec != 'both':
            return 5
        if self.sched == 5:
            return 3                 # eager steps of the dual-graph schedule use plain stream edges
        return 3 if self.sched == 3 else 1

    # ------------------------------------------------------------------- hipGraph
    def _launch_sequence(self, allreduce=None):
        self.fuse_bwd = True
        try:
            self.draw_noise(bump=False)
            self.forward()
            self.backward()
            if allreduce is not None:
                allreduce(self.arena.xchg)
            self.optimizer_step()
        finally:
            self.fuse_bwd = False

    def capture(self, split_for_allreduce=False, allreduce=None):
        """Capture the train step (Philox noise + forward + backward + Adam: ~100 launches)
        into hipGraph(s) for the current batch structure.  With ``split_for_allreduce`` the
        step is captured as two graphs so that an (uncaptured) RCCL all-reduce of the
        gradient arena can run between backward and Adam; ``split_for_allreduce='captured'`` with
        ``allreduce``: the collective is captured INTO the step's graph (RCCL supports stream capture),
        no graph boundary, no host-side launch of the exchange."""
        self._captured_allreduce = allreduce if split_for_allreduce == 'captured' else None
        assert self.plan is not None, 'set_batch first'
        self.join_side()
        if self.plan.DZMMD is not None and split_for_allreduce:
            raise NotImplementedError('use_MMD: the model-level MMD penalty is a cross-row term (every row of a nuisance '
                                      'class against every other row): it cannot be sharded over ranks')
        self.training = True
        self.plan.set_beta(self.beta_pert())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):         # warm-up on a side stream (loads code objects)
            self.draw_noise(bump=False)       # (the Philox counter is NOT advanced: capturing leaves the stream
            self._rng_pending = 0             #  of draw events exactly where an eager step would find it)
            self.forward()
            self.backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graphs = []
        self._side_graph = None
        # two flag-ordered graphs only for the latency-bound steps: once the decoder products alone fill the chip many
        # times over (wide configuration) the side chain's small kernels, squeezed in between the resident GEMM
        # workgroups of a second queue, cost more than they hide (36.7 ms dual, 35.9 ms as one graph with a fork/join)
        dual = self.sched == 5 and self._dual_capable() and self._latency_bound() and self._flags_usable()
        if dual and not self.cfg.has_y and split_for_allreduce:
            # PVAE's side chain is the step's tail only (optimiser half, loss scalars, noise): under a gradient exchange
            # the optimiser half cannot move there and the rest does not pay for the second graph (one-rank RCCL, cfg 1:
            # 0.1848 -> 0.1913 split, 0.1695 -> 0.1737 captured)
            dual = False
        self._split_capture = bool(split_for_allreduce)
        self._split_kind = split_for_allreduce          # False | True (two graphs) | 'overlap' | 'captured'
        cfg = self.cfg
        # (under a CAPTURED exchange too: the side chain then draws behind the join and sweeps its half behind the collective)
        cap_fork = bool(dual and split_for_allreduce == 'captured' and self._cap_fork(5, 'captured', self._side_adam_layout()[0]))
        self.noise_ahead = bool(dual and (split_for_allreduce in (False, True) or cap_fork) and self._late_ok()
                                and T.get('noise_ahead'))
        self._noise_stale = True
        if dual:
            self._rec = 'main'
        # no garbage collection while a stream is capturing: a collected cycle may own device or pinned memory,
        # events or graphs of a retired engine, and releasing those calls HIP functions that are illegal
        # during (global-mode) capture -- the process aborts
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        # a chip-filling step (wide configuration) is captured on ONE stream: its side chain is 0.5 ms of small launches
        # next to 31 ms of products that want every CU; as a graph branch they squeeze in between the resident GEMM
        # workgroups of the other queue and cost more than they hide (measured, round 4: 32.0 ms with the fork/join,
        # 31.5 ms in order on one stream)
        branch_on = self.branch.on
        self._late_fork = False
        if branch_on and not dual and not self._latency_bound() and T.get('wide_single') == 2:
            # (round 6) ... or as a branch forked LATE: behind the decoder heads' product, next to the HBM-bound NLL row pass
            # -- the one stretch of the main chain that leaves the matrix pipes idle -- and joined behind the decoder's
            # backward products (the side chain's 0.4 ms of small launches are through long before)
            self._late_fork = True
        elif branch_on and not dual and not self._latency_bound() and T.get('wide_single'):
            self.branch.on = False
        try:
            self._capture_main(split_for_allreduce)
            if dual:
                self._rec = 'side'
                self.sync_side_counters()
                self.flag_side.wait_stream(torch.cuda.current_stream())
                gs = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gs, stream=self.flag_side):
                    self.fuse_bwd = True
                    try:
                        self.forward()
                        self.backward()
                    finally:
                        self.fuse_bwd = False
                self._side_graph = gs
        finally:
            self._rec = 'both'
            self.branch.on = branch_on
            self._late_fork = False
            if gc_was_on:
                gc.enable()
        self._graph_key = self.plan.key
        self._graph_feed = self.plan.live_feed
        self._graph_mmd_sig = getattr(self.plan, 'mmd_sig', None)
        return self

    # ------------------------------------------------------------ CU partition
    def _latency_bound(self):
        """the step's big products do not fill the chip many times over (<= 4 M elements of decoder output)"""
        return self.plan is not None and self.plan.DPX.shape[0] * self.plan.DPX.shape[1] <= (4 << 20)

    def _partition_applicable(self):
        # only for latency-bound steps: once the decoder products alone fill the chip many times over
        # (wide configuration) the main chain needs every CU (measured: 52 ms -> 66 ms/step when masked)
        small = self._latency_bound()
        # (a model without a classifier -- PVAE's tail-only side chain -- only once its step IS captured as two graphs:
        # under a gradient exchange it is not, and a masked main stream would only cost it CUs)
        if not self.cfg.has_y and self._side_graph is None:
            return False
        return bool(self.sched == 5 and self._dual_capable() and small and
                    int(os.environ.get('DRVAE_SIDE_CUS', '64')) > 0)

    def _part_streams(self, n_side):
        """(main, side) CU-masked streams reserving ``n_side`` CUs for the side chain (cached)"""
        cache = _PARTITION_STREAMS.setdefault(torch.device(self.dev).index or 0, {})   # per device, process-wide:
        n_side = int(n_side)
        if (n_side, bool(T.get('part_xcd'))) not in cache:                    # masked streams own hardware queues, so engines share them
            n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count
            words = (n_cu + 31) // 32
            side_bits, all_bits = [0] * words, [0] * words
            # bit i of a CU mask = XCD i % 8, shader engine (i / 8) % 4 of it, CU i / 32 of that engine (tools/cumask_probe.hip).
            # 'part_xcd': the reserve is WHOLE XCDs (the last n_side / 32 of the eight) -- the side chain then has L2s of its
            # own, the main chain keeps six (five, ...) undivided ones; default: the first n_side bits = n_side / 32 CUs of
            # every shader engine of every XCD
            by_xcd = bool(T.get('part_xcd')) and n_side % 32 == 0 and 0 < n_side < n_cu and n_cu == 256
            for i in range(n_cu):
                all_bits[i // 32] |= 1 << (i % 32)
                if (i % 8 >= 8 - n_side // 32) if by_xcd else (i < min(n_side, n_cu - 1)):
                    side_bits[i // 32] |= 1 << (i % 32)
            main_bits = [a & ~b for a, b in zip(all_bits, side_bits)]
            try:
                pair = (_masked_stream(main_bits, self.dev), _masked_stream(side_bits, self.dev))
                with torch.cuda.stream(pair[0]):
                    if not self._probe(pair[1]):         # must sit on different hardware queues
                        pair = None
            except (OSError, RuntimeError, AttributeError) as e:      # runtime without CU masking: plain streams
                import warnings
                warnings.warn('drvae_amd: CU partition unavailable (%s)' % e)
                pair = None
            cache[(n_side, bool(T.get('part_xcd')))] = pair
        return cache[(n_side, bool(T.get('part_xcd')))]

    def partition(self, n_side=None):
        """Context manager: run the train step with the GPU's compute units split between the two
        launch chains -- the side chain (many small launches) on ``n_side`` reserved CUs, the main
        chain (the big GEMMs) on the rest -- via CU-masked HIP streams.  Inside the context the
        masked main stream is the current stream, so everything the caller enqueues (batch feed,
        loss accumulation, replays) is ordered with the step; on exit the outer stream waits for it.
        Measured on MI355X (cfg 2): 64 reserved CUs take the step from 0.264 to 0.240 ms; without the
        reservation the side chain's small kernels queue behind the GEMM workgroups and the main
        chain waits ~32 us per step at the join.  ``tune_partition()`` picks the split by timing.
        No-op when not applicable, disabled (DRVAE_SIDE_CUS=0) or off-GPU."""
        import contextlib
        if not self._partition_applicable():
            return contextlib.nullcontext()
        if n_side is None:
            n_side = getattr(self, '_side_cus', None) or int(os.environ.get('DRVAE_SIDE_CUS', '64'))
        pair = self._part_streams(n_side)
        if pair is None:
            return contextlib.nullcontext()
        main, side = pair

        @contextlib.contextmanager
        def ctx():
            outer = torch.cuda.current_stream()
            prev = self._flag_side
            main.wait_stream(outer)
            side.wait_stream(outer)
            self._flag_side = side
            try:
                with torch.cuda.stream(main):
                    yield self
            finally:
                outer.wait_stream(main)
                outer.wait_stream(side)
                self._flag_side = prev
        return ctx()

    def tune_partition(self, candidates=(32, 64, 96, 128), steps=24):
        """Pick the CU split of ``partition()`` by timing replays of the captured step (the best split
        depends on how the two chains balance, i.e. on the model and on the individual GPU).  Runs on a
        scratch copy of the training state: parameters, Adam moments and all device counters are restored
        afterwards.  Returns the chosen number of reserved CUs (None when partitioning does not apply).
        Candidates are multiples of 32: bit i of a CU mask is a CU of shader engine i % 32 (8 XCDs x 4), and the dispatcher
        hands every shader engine the same share of a grid -- a reserve that gives some engines one CU less makes those
        the pace of the whole side chain (cfg 2: 56 reserved CUs 0.249 ms, 64: 0.1975, 72: 0.2125; every 8th / 4th / 2nd
        bit instead of the first 64: 0.236; masks that overlap -- CUs open to both chains: 0.216-0.258)."""
        if not (self._partition_applicable() and self._side_graph is not None and len(self._graphs) == 1) or \
                'DRVAE_SIDE_CUS' in os.environ:       # (multi-rank: split graphs need the exchange; keep the default)
            return None
        a = self.arena
        keep = [t.clone() for t in (a.param, a.exp_avg, a.exp_avg_sq, self.step_dev, self.side_ctr, self.side_t, self.rng_ctr,
                                    self.flags)]
        iters = self.iters
        best = (None, float('inf'))
        for n in candidates:
            if self._part_streams(n) is None:
                continue
            with self.partition(n):
                for _ in range(4):
                    self.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    self.replay()
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1)
            if t < best[1]:
                best = (n, t)
        torch.cuda.synchronize()
        for dst, src in zip((a.param, a.exp_avg, a.exp_avg_sq, self.step_dev, self.side_ctr, self.side_t, self.rng_ctr,
                             self.flags), keep):
            dst.copy_(src)
        self.iters = iters
        self._noise_stale = True
        self.plan.set_beta(self.beta_pert())
        torch.cuda.synchronize()
        self._side_cus = best[0]
        return best[0]

    @property
    def flag_side(self):
        """the stream the side-chain graph of the dual-graph schedule is launched on: CU-masked inside
        ``partition()``, otherwise a plain stream that ``_flags_usable`` has verified to sit on a
        different hardware queue than the launching stream."""
        if self._flag_side is None:
            self._flag_side = torch.cuda.Stream(device=self.dev)
        return self._flag_side

    def _probe(self, side):
        """park a short wait on the launching stream, publish from ``side``: a timeout means the two
        streams share a hardware queue (a parked wait kernel blocks everything behind it in its queue)"""
        probe = torch.zeros(4, dtype=torch.int32, device=self.dev)      # flag, counter, error, ticks
        torch.cuda.synchronize()
        K.flag_wait(probe[0:1], probe[1:2], probe[2:4], add=1, max_spins=20000)
        with torch.cuda.stream(side):
            K.flag_publish(probe[0:1], probe[1:2], 1)
        torch.cuda.synchronize()
        return int(probe[2]) == 0

    def _flags_usable(self):
        """Device-flag ordering needs the two streams on DIFFERENT hardware queues.  HIP multiplexes its
        streams onto a few queues, so probe candidates until one qualifies (the CU-masked stream of
        ``partition()`` has a queue of its own); otherwise fall back to graph edges.  (High-priority
        streams are avoided on purpose: with one in the process, captured fork/joins ran 2.4x slower.)"""
        if getattr(self, '_flags_ok', None) is None:
            ok = self._probe(self.flag_side)
            tries = 0
            while not ok and not getattr(self, '_part', None) and tries < 8:
                self._flag_side = torch.cuda.Stream(device=self.dev)
                ok = self._probe(self._flag_side)
                tries += 1
            self._flags_ok = ok
            if not ok:
                import warnings
                warnings.warn('drvae_amd: no side stream on a hardware queue of its own; using graph edges')
        return self._flags_ok

    def check_sync(self):
        """raise if a device-side wait of the dual-graph schedule ever timed out (results would be stale)"""
        self.join_side()
        if int(self.sync_err[0::2].abs().sum()) != 0:
            self._raise_sync(self.sync_err.cpu().tolist())

    @staticmethod
    def _raise_sync(words):
        sites = [i for i, v in enumerate(words[0::2]) if v != 0]
        raise RuntimeError('drvae_amd: a device-side chain wait timed out (main / side stream ordering; wait site(s) %s). '
                           'From that step on the loss scalars are NaN and the optimiser leaves the parameters '
                           'untouched: the state is that of the last good step -- except when the timed-out wait was the '
                           'optimiser gate itself (site 3): that one step is applied to every parameter but the gated '
                           'slice (the classifier head).  Restore from the last checkpoint.' % sites)

    def _poll_sync(self):
        """Called once per replayed step.  Every ``SYNC_POLL`` steps the sticky error words of the device-side
        waits are copied to pinned host memory asynchronously; the copy issued one period earlier is checked
        first (waiting for it bounds how far the host runs ahead to two periods of queued steps, so the device
        never idles on it).  A timed-out wait therefore raises within two periods even in a ``replay()`` loop
        that never reads the losses."""
        self._since_poll = getattr(self, '_since_poll', 0) + 1
        if self._since_poll < SYNC_POLL:
            return
        self._since_poll = 0
        if getattr(self, '_sync_host', None) is None:
            # pinned landing buffers are pooled for the life of the process: returning one to torch's host
            # allocator queries its events, which is illegal while ANY stream is capturing -- and garbage
            # collection may run in the middle of a later capture
            if _PINNED_POOL:
                self._sync_host = _PINNED_POOL.pop()
            else:
                self._sync_host = torch.zeros(self.sync_err.numel(), dtype=torch.int32).pin_memory()
                ctypes.pythonapi.Py_IncRef(ctypes.py_object(self._sync_host))   # never deallocated, not even at exit
            weakref.finalize(self, _PINNED_POOL.append, self._sync_host).atexit = False
            self._sync_event = None
        if self._sync_event is not None:
            self._sync_event.synchronize()
            words = self._sync_host.tolist()
            if any(words[0::2]):
                self._raise_sync(words)
        self._sync_host.copy_(self.sync_err, non_blocking=True)
        self._sync_event = torch.cuda.Event()
        self._sync_event.record()

    def _capture_main(self, split_for_allreduce):
        if split_for_allreduce == 'overlap' and self.arena.late_end < self.arena.xchg.numel():
            # three graphs: [noise .. decoder backward] | [rest of backward] | [Adam]; the exchange of the
            # decoder block is launched between the first two and travels while the second runs
            ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(device=self.dev)
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                def split():
                    ga.capture_end()
                    gb.capture_begin()
                self._after_decoder_bwd = split
                self.fuse_bwd = True
                try:
                    ga.capture_begin()
                    self.draw_noise(bump=False)
                    self.forward()
                    self.backward()
                    gb.capture_end()
                finally:
                    self.fuse_bwd = False
                    self._after_decoder_bwd = None
                gc.capture_begin()
                self.optimizer_step()
                gc.capture_end()
            torch.cuda.current_stream().wait_stream(cap)
            self._graphs = [ga, gb, gc]
        elif split_for_allreduce and split_for_allreduce != 'captured':
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self.fuse_bwd = True
                try:
                    self.draw_noise(bump=False)
                    self.forward()
                    self.backward()
                finally:
                    self.fuse_bwd = False
            with torch.cuda.graph(g2):
                self.optimizer_step()
            self._graphs = [g1, g2]
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._launch_sequence(allreduce=getattr(self, '_captured_allreduce', None))
            self._graphs = [g]

    def replay(self, allreduce=None):
        """One captured train step.  New data is fed by copying into plan.x1 / plan.x2 in place."""
        assert self._graph_key == self.plan.key, 'batch structure changed: capture again'
        assert self._graph_feed is self.plan.live_feed, 'input source changed (epoch feed <-> explicit batch): capture again'
        assert getattr(self, '_graph_mmd_sig', None) == getattr(self.plan, 'mmd_sig', None), \
            'use_MMD: the nuisance classes of the batch changed (the penalty compares row sets): capture again'
        self.plan.set_beta(self.beta_pert())      # 0.01 on iteration 0, 1.0 afterwards (device-side coefficients)
        if self.noise_ahead and self._noise_stale:        # first replay (or an eager draw since): this step's noise
            K.fill_normal_rows(self.plan.noise, self.plan.noise_desc, self.seed, self.rng_ctr)
            self._noise_stale = False
        # (the host needs ~60 us per graph launch: with the side chain's graph launched first -- it only parks on the z1 flag -- the
        # step's first kernels started a launch later whenever the device had run dry: after a host sync, i.e. at the head of every
        # timed region and of every epoch; 20 steps behind a sync: 0.1905 -> 0.188 ms per step, steady state 0.183 either way)
        main_first = self._side_graph is not None and T.get('main_first')
        if self._side_graph is not None and not main_first:         # first: its wait kernel is parked before the main chain publishes
            with torch.cuda.stream(self.flag_side):
                self._side_graph.replay()
        self._graphs[0].replay()
        if main_first:
            with torch.cuda.stream(self.flag_side):
                self._side_graph.replay()
        if len(self._graphs) == 3:               # overlapped exchange: ``allreduce`` has start()/finish()
            a = self.arena
            w_early = allreduce.start(a.xchg[a.late_end:])
            self._graphs[1].replay()
            w_late = allreduce.start(a.xchg[:a.late_end])
            allreduce.finish(w_early)
            allreduce.finish(w_late)
            self._graphs[2].replay()
        elif len(self._graphs) == 2:
            if self._side_graph is not None:
                # the side chain's work behind the join (leaf gradients, loss scalars) must be final before the
                # exchange reads the buffer: order the side stream in front of it
                torch.cuda.current_stream().wait_stream(self.flag_side)
            if allreduce is not None:
                allreduce(self.arena.xchg)
            self._graphs[1].replay()
        self.iters += 1
        if self._side_graph is not None:
            self._poll_sync()
