"""Data parallelism for the train step: one process per GPU, the minibatch rows sharded
across ranks, parameters + Adam state replicated, ONE exchange per step -- a sum
all-reduce (RCCL over xGMI; ``torch.distributed`` backend "nccl" is RCCL on ROCm) of the
flat gradient arena, whose tail also carries the loss scalars.  The reference has no
distributed code; semantics are defined in SURVEY.md 8(e): every rank normalises its
row sums by the GLOBAL counts (N_total, N_pairs, N_labeled) so that summed shard
gradients equal the gradient of the concatenated batch, and all ranks then apply the
identical fused Adam update."""
import os

import numpy as np
import torch
import torch.distributed as dist


def force_dp():
    """DRVAE_FORCE_DP=1: run the data-parallel step path -- process group, split graphs, RCCL launches between
    the graph replays -- in a single process (a one-rank communicator): a functional check of everything but
    the peers on a one-GPU box"""
    return os.environ.get('DRVAE_FORCE_DP') == '1'


def _active():
    return dist.is_initialized() and (dist.get_world_size() > 1 or force_dp())


def _pg_timeout():
    """bound of every collective (DRVAE_DIST_TIMEOUT seconds, default 180): a rank that never arrives must end the job
    with a non-zero exit code, not hang it -- gloo raises in the blocked call, RCCL's watchdog thread tears the process
    down (TORCH_NCCL_ASYNC_ERROR_HANDLING=1)"""
    import datetime
    os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')
    return datetime.timedelta(seconds=float(os.environ.get('DRVAE_DIST_TIMEOUT', '180')))


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world_size, local_rank); a no-op (0, 1, 0) for single-process runs."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and force_dp():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if not dist.is_initialized():
            if torch.cuda.is_available():
                torch.cuda.set_device(0)
            dist.init_process_group(backend=backend or os.environ.get('DRVAE_DIST_BACKEND') or
                                    ('nccl' if torch.cuda.is_available() else 'gloo'), rank=0, world_size=1,
                                    timeout=_pg_timeout())
        return 0, 1, 0
    if world <= 1:
        return 0, 1, 0
    rank, local = int(os.environ['RANK']), int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:      # DRVAE_DIST_BACKEND=gloo: functional test of the multi-rank path on one GPU
        backend = os.environ.get('DRVAE_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=_pg_timeout())
    return rank, world, local


def allreduce_sum(flat):
    """In-place sum over ranks of a flat fp32 tensor (the gradient arena)."""
    if _active():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


class OverlappedAllReduce:
    """Two-piece gradient exchange overlapped with the backward pass: ``start`` launches an
    asynchronous sum all-reduce (RCCL runs it on its own stream once the work enqueued so far on the
    current stream is done), ``finish`` makes the current stream wait for it.  xGMI is point-to-point
    and a 9 MB all-reduce is latency-bound, so the decoder block (5 MB, final two thirds into the
    step) travels while the encoder backward still runs."""

    def start(self, flat):
        if _active():
            return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        return None

    def finish(self, work):
        if work is not None:
            work.wait()


def global_counts(has_x2, has_y, kind='drvae', semi_supervised=True, local=False):
    """(N_total, N_pairs, N_labeled) over all ranks (tiny host-side all-reduce; the flags
    come from the host data pipeline).  ``local``: the flags already are the whole set's (no collective)."""
    hx = np.asarray(has_x2).astype(bool).reshape(-1)
    hy = np.asarray(has_y).astype(bool).reshape(-1)
    n_tot = int(hy.sum()) if (kind == 'vfae' and not semi_supervised) else len(hy)
    c = torch.tensor([n_tot, int(hx.sum()) if kind != 'vfae' else 0, int(hy.sum()) if kind != 'pvae' else 0],
                     dtype=torch.float64)
    if not local and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == 'nccl':
            c = c.cuda()
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        c = c.cpu()
    return tuple(int(v) for v in c.tolist())


def shard_rows(n_rows, rank, world):
    """contiguous row range [lo, hi) of this rank (SURVEY.md 8(e) partitioning)."""
    per = n_rows // world
    assert per * world == n_rows, 'global batch must divide evenly over ranks'
    return rank * per, (rank + 1) * per


def broadcast(t, src=0):
    """rank ``src``'s tensor to every rank, in place (a CPU tensor travels through the device under RCCL)"""
    if not _active():
        return t
    if dist.get_backend() == 'nccl' and not t.is_cuda:
        d = t.cuda()
        dist.broadcast(d, src=src)
        t.copy_(d.cpu())
    else:
        dist.broadcast(t, src=src)
    return t


def broadcast_int(v, src=0):
    return int(broadcast(torch.tensor([int(v)], dtype=torch.int64), src)[0])


def broadcast_params(arena):
    """make every rank start from rank 0's parameters"""
    broadcast(arena.param)
