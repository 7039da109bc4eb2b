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


def epoch_schedule(n_rows, world, b):
    """The steps of one data-parallel epoch as ``[(rows per rank, global offset, global rows)]``: every full global
    batch of ``world * b`` rows, then -- the reference keeps the last partial batch (``drop_last=False``,
    sc/clustering/dataloader.py:71) -- the tail split evenly over the ranks (``tail // world`` rows each; at most
    ``world - 1`` rows of the permutation stay unused, none when world == 1; a tail of fewer than 2 rows per rank is
    dropped because training-mode BatchNorm needs two).  Rank r steps rows
    ``perm[offset + r * rows : offset + (r + 1) * rows]`` of each entry."""
    steps, off = [], 0
    for _ in range(n_rows // (world * b)):
        steps.append((b, off, world * b))
        off += world * b
    tail = (n_rows - off) // world
    if tail >= 2 or (world == 1 and tail == 1):
        steps.append((tail, off, world * tail))
    return steps


def broadcast_from_rank0(values, device, group=None):
    """A short list of floats, rank 0's copy on every rank (float64, exact)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64)
    if dist.get_backend(group) == "nccl":
        t = t.to(device)
    dist.broadcast(t, src=0, group=group)
    return t.cpu().tolist()


def broadcast_tensor_from_rank0(t, device, group=None):
    """An int64 / float CPU tensor (e.g. the epoch permutation): rank 0's values on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    buf = t.to(device) if dist.get_backend(group) == "nccl" else t.clone()
    dist.broadcast(buf, src=0, group=group)
    return buf.cpu()


def any_rank(flag, device, group=None):
    """True on every rank iff ``flag`` is true on at least one (one tiny MAX all-reduce; every rank must call it)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.cpu()[0]))
