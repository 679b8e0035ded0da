"""Developer switches of the train step's launch schedule.  NOT part of the user surface: every default is the
measured best on MI355X ("Step scheduling"), and the alternatives exist so that tools/ (A/B runs, the
soak test) can still reach them.  One variable carries them all::

    DRVAE_TUNE="sched=3,fold_tail=1" python bench.py

User-facing environment variables (the whole list): DRVAE_HIP_LIB (another build of the library), DRVAE_SIDE_CUS
(compute units reserved for the side chain; 0 = no partition), DRVAE_WAIT_SPINS (bound of a device-side wait),
DRVAE_DIST_BACKEND (torch.distributed backend of the data-parallel step; default nccl = RCCL on a GPU),
DRVAE_FORCE_DP=1 (the multi-rank step path over a one-rank communicator: functional check on a one-GPU box)."""
import os

DEFAULTS = {
    'sched': 5,          # 1 graph fork/join per pass | 3 one fork/join per step | 5 two graphs ordered by device flags
    'late_leaf': 1,      # classifier dW behind the side chain's publish (gated optimiser sweep)
    'side_adam': 1,      # the decoder-heads half of the optimiser sweep on the side chain
    'fold_join': 1,      # the join parks on its first consumer instead of a launch of its own
    'fuse_heads': 1,     # samples / NLL forward+backward in the epilogue of the heads' GEMM (dv_gemm_heads)
    'fold_waits': 1,     # the side chain's second wait rides on the row kernel behind it
    'fold_tail': 5,      # bit mask: 1 / 4 the side chain's publishes ride on the next launch, 2 the noise draw parks itself
    'klz2_main': -1,     # pairs' KL rows on the main chain: -1 = by plan kind (structured: yes, universal: no)
    'fprop_tail': 1,     # fprop KL rows + z1 term's backward on the classifier-head launch
    'fprop_heads': 1,    # z1 samples copied into the fprop input by the encoder heads' epilogue
    'clf_small': 1,      # single-Linear classifier with <= 8 classes as wave-per-row kernels
    'wbranch': 0,        # weight gradients on a third graph branch (measured slower)
    'noise_ahead': 1,    # the side chain draws the NEXT step's noise behind the join
    'raw_heads': 1,      # chip-filling decoder heads (train step) as a plain product, finished by the NLL row pass
    'tail_gate': 1,      # the side chain's tail is awaited by the NEXT step's first launch (0: by this step's optimiser launch)
    'concurrent': 1,     # side chain at all (0: one stream)
    'part_xcd': 0,       # the side chain's CU reserve as whole XCDs (1) instead of n/32 CUs of every shader engine (0)
    'wide_single': 2,    # chip-filling steps (wide configuration): 1 = captured on one stream, no fork/join; 2 = the side chain as a branch forked LATE, next to the HBM-bound NLL row pass (cfg 5: 31.09-31.15 -> 30.87-30.89 ms); 0 = forked at the start of the step (31.2-31.4)
    'nll_cs': 1,         # chip-filling heads: their bias gradient folded into the NLL row pass (no column-sum pass of its own; 2: buffers at any size -- tests, with raw_heads=2)
    'klq_epi': 1,        # the z3 term's backward (KL + sample path of q(z3|z1,y)) in the epilogue of the data-gradient product in front of it
    'kl_pair': 1,        # PVAE's two sets of KL rows (prior term, pairs' term) as one launch on its main chain
    'main_first': 1,     # replay(): the main chain's graph is launched before the side chain's (the step's first kernels start one graph launch earlier after a host sync; steady state unchanged)
    'pvae_tail': 1,      # PVAE (no classifier): the dual-graph schedule with a side chain that is only the step's tail (heads' optimiser half, loss scalars, next noise)
    'mmd_explicit': 1,   # model-level MMD penalty (use_s extension, rbf_fourier / identity kernels) as explicit launch lists, no autograd inside the step
    'dp_fork': 1,        # captured gradient exchange: the side chain draws the next step's noise behind the join (as in the single-GPU step)
    'sync_poll': 64,     # replays between two polls of the sticky wait-error words
}


def _parse():
    out = dict(DEFAULTS)
    for kv in filter(None, os.environ.get('DRVAE_TUNE', '').split(',')):
        k, _, v = kv.partition('=')
        k = k.strip()
        if k not in DEFAULTS:
            raise ValueError('DRVAE_TUNE: unknown switch %r (known: %s)' % (k, ', '.join(sorted(DEFAULTS))))
        out[k] = int(v)
    return out


_VALUES = None


def get(name):
    global _VALUES
    if _VALUES is None:
        _VALUES = _parse()
    return _VALUES[name]


def reload():
    """re-read DRVAE_TUNE (tests that change the environment)"""
    global _VALUES
    _VALUES = None
