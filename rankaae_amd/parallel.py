"""Data-parallel helpers: one process per GPU, ``torch.distributed`` over RCCL (backend "nccl" on
ROCm) -- replaces the reference's ipyparallel trial farm (``sc/cmd/train_sc.py:25-45``) for the
training path, per the north star.  The same code runs over ``gloo`` on CPU in the tests.

Sharding: every rank holds the whole training split and the SAME epoch permutation; global batch i
is rows ``perm[i*W*b : (i+1)*W*b]`` and rank r steps rows ``[r*b, (r+1)*b)`` of it.  Per phase the
flat gradient arena of the optimizer's range is averaged with ONE all-reduce;
BatchNorm statistics and the rank-loss pairs stay per replica (DDP semantics).
"""
import torch
import torch.distributed as dist


def shard_rows(perm, rank, world, b, i):
    """Rows of global batch ``i`` that rank ``rank`` steps."""
    start = i * world * b + rank * b
    return perm[start:start + b]


def cursor_params(rank, world, b):
    """``(start, stride)`` for ``StepEngine.set_epoch``: the device row cursor advances by ``stride``
    per step and the gather reads ``perm[start + i*stride : +b]``."""
    return rank * b, world * b


def full_global_batches(n_rows, world, b):
    return n_rows // (world * b)


def allreduce_mean_(buf, group=None):
    """In-place mean over ranks: ``ReduceOp.AVG`` on RCCL (one kernel), SUM + scale on gloo (which has no AVG)."""
    world = dist.get_world_size(group)
    if world == 1:
        return buf
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        buf.mul_(1.0 / world)
    return buf
