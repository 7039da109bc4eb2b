"""An RCCL communicator of our own (ctypes), so that the per-phase gradient all-reduce can be CAPTURED into the
step's hipGraph instead of cutting the graph five times (the collectives are latency-bound,
29/10/28/28/18 KB for the conv networks).

``torch.distributed`` stays the bootstrap (the 128-byte unique id travels through it) and the fallback: the
communicator is only used after a self-test -- an eager all-reduce and a captured-and-replayed one with known
answers -- passed on EVERY rank (agreement through the torch process group); otherwise the engine keeps the
segmented path (``StepEngine._collective``).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_FLOAT32, NCCL_FLOAT64, NCCL_SUM, NCCL_AVG = 7, 8, 0, 4


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * 128)]      # not c_char: ctypes would cut the id at its first NUL byte


def _load():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")     # the copy torch itself uses
    lib = C.CDLL(path)
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclAllGather, lib.ncclCommDestroy):
        f.restype = C.c_int
    return lib


def agree(ok, device, group=None):
    """True only when ``ok`` on EVERY rank of ``group`` (one MIN all-reduce on the torch process group; every rank
    must call it exactly once per decision)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = bool(int(flag) == 1)
    return bool(ok)


class GraphAllReduce:
    """Mean all-reduce of fp32 device buffers on the CURRENT torch stream; legal under hipGraph capture."""

    def __init__(self, rank, world, device, group=None):
        self.rank, self.world, self.device, self.group = rank, world, device, group
        # every step of the bootstrap is followed by an agreement of ALL ranks (one MIN all-reduce on the torch
        # process group), so that a rank that failed never leaves the others inside a collective it skipped: rank 0
        # always enters the unique-id broadcast (with a zero id when it could not make one), and the communicator
        # is only initialised -- collectively, it blocks until every rank has joined -- once every rank has the
        # library loaded and a valid id
        self.lib, uid, ok = None, _UniqueId(), True
        try:
            self.lib = _load()
            if rank == 0:
                self._check(self.lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        except Exception:                                     # noqa: BLE001
            ok = False
        box = [(C.string_at(C.byref(uid), 128) if ok else bytes(128)) if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        ok = ok and any(box[0])
        self.comm = C.c_void_p()
        if not agree(ok, device, group):
            raise RuntimeError("RCCL bootstrap failed on at least one rank")
        C.memmove(C.byref(uid), box[0], 128)
        with torch.cuda.device(device):
            self._check(self.lib.ncclCommInitRank(C.byref(self.comm), world, uid, rank), "ncclCommInitRank")

    def _check(self, code, what):
        if code != 0:
            raise RuntimeError(f"{what}: {self.lib.ncclGetErrorString(code).decode()}")

    def mean_(self, buf):
        assert buf.dtype == torch.float32 and buf.is_contiguous() and buf.device == self.device
        stream = torch.cuda.current_stream().cuda_stream
        self._check(self.lib.ncclAllReduce(C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), buf.numel(),
                                           NCCL_FLOAT32, NCCL_AVG, self.comm, C.c_void_p(stream)), "ncclAllReduce")
        return buf

    def sum_(self, buf):
        """In-place SUM of a float64 device buffer (the rank loss's pair totals under global pairs)."""
        assert buf.dtype == torch.float64 and buf.is_contiguous() and buf.device == self.device
        stream = torch.cuda.current_stream().cuda_stream
        self._check(self.lib.ncclAllReduce(C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), buf.numel(),
                                           NCCL_FLOAT64, NCCL_SUM, self.comm, C.c_void_p(stream)), "ncclAllReduce")
        return buf

    def gather_(self, src, dst):
        """``dst[r * n : (r + 1) * n] = src`` of rank r (fp32, n = src.numel())."""
        assert src.dtype == dst.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous()
        assert dst.numel() == self.world * src.numel()
        stream = torch.cuda.current_stream().cuda_stream
        self._check(self.lib.ncclAllGather(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.numel(),
                                           NCCL_FLOAT32, self.comm, C.c_void_p(stream)), "ncclAllGather")
        return dst

    def self_test(self):
        """Known-answer test, eager and captured; returns True only when every rank passed."""
        from . import ops
        ok = True
        try:
            side = torch.cuda.Stream(device=self.device)      # the legacy default stream cannot capture
            torch.cuda.synchronize(self.device)
            with torch.cuda.stream(side):
                want = (self.world + 1) / 2.0                 # mean of rank + 1 over ranks
                x = torch.full((4096,), float(self.rank + 1), device=self.device)
                self.mean_(x)
                side.synchronize()
                ok = bool(torch.allclose(x, torch.full_like(x, want)))
                y = torch.empty(4096, device=self.device)
                side.synchronize()
                g = ops.Graph()
                g.begin()
                y.fill_(float(self.rank + 1))
                self.mean_(y)
                g.end()
                for _ in range(2):
                    g.launch()
                side.synchronize()
                ok = ok and bool(torch.allclose(y, torch.full_like(y, want)))
                # the two collectives of the global-pairs rank loss: all-gather (fp32) and sum (fp64), eager + captured
                src = torch.full((16,), float(self.rank + 1), device=self.device)
                dst = torch.zeros(16 * self.world, device=self.device)
                tot = torch.full((8,), float(self.rank + 1), dtype=torch.float64, device=self.device)
                self.gather_(src, dst)
                self.sum_(tot)
                side.synchronize()
                want_g = torch.arange(1, self.world + 1, device=self.device, dtype=torch.float32).repeat_interleave(16)
                ok = ok and bool(torch.equal(dst, want_g)) and bool(torch.allclose(tot, torch.full_like(tot, self.world * (self.world + 1) / 2.0)))
                g2 = ops.Graph()
                g2.begin()
                dst.zero_()
                tot.fill_(float(self.rank + 1))
                self.gather_(src, dst)
                self.sum_(tot)
                g2.end()
                g2.launch()
                side.synchronize()
                ok = ok and bool(torch.equal(dst, want_g)) and bool(torch.allclose(tot, torch.full_like(tot, self.world * (self.world + 1) / 2.0)))
        except Exception:                                     # noqa: BLE001 -- any failure means "fall back"
            ok = False
        return agree(ok, self.device, self.group)

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
