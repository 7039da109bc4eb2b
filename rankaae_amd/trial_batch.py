"""Independent trials batched into one launch sequence.

The reference's real workload is several small INDEPENDENT trainings of one configuration (``trials: 8`` in
``example/fix_config.yaml``, mapped over ipyparallel engines by ``sc/cmd/train_sc.py:127-143``).  A 256-row step of the
dense networks launches ~100 kernels of 16-64 workgroups each: one trial leaves most of the 256 CUs idle, and streams
of concurrent trials saturate at the four hardware queues (2.8-3.5x).  A ``TrialBatch`` steps T engines with ONE
launch sequence: every kernel of the step runs once with ``gridDim.z = T``, plane z working on trial z's buffers
(``include/rankaae_hip.h``: raae_record_* / raae_multi_*).  Each trial's arithmetic is the kernel body it runs alone:
its weights are bit for bit those of the same trial stepped by itself (tests/test_engine_gpu.py).

The first step of every batch shape is run by each engine on its own (eagerly: that is where an engine records its
gradient-slab tables) while the library logs its launches; from the second step on the batch replays one captured
hipGraph.  All engines of a batch share one HIP stream.  Supported: both networks at batches below 1024 rows (the
launch-bound kernel instances have the batched form; a step that meets a large-batch instance or a per-layer kernel is
refused with an error, never run partially), ``rng_mode: philox``, fp32, one GPU.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import check


class TrialBatch:
    def __init__(self, engines):
        assert len(engines) >= 1
        e0 = engines[0]
        for e in engines:
            if e.rng_mode != "philox" or e.world_size != 1 or e.bf16:
                raise ValueError("a TrialBatch takes fp32 engines with the device RNG (rng_mode: philox) on one GPU")
            if e.tile_mult != engines[0].tile_mult:
                raise ValueError("the engines of a TrialBatch must share one tile_rows_mult")
            if e.stream is not e0.stream:
                raise ValueError("the engines of a TrialBatch must share one HIP stream (StepEngine(..., stream=s))")
            if not e.cfg.get("fused_step_begin", True) or not e.cfg.get("fused_discriminator", True):
                raise ValueError("a TrialBatch needs the fused step head and the fused discriminator")
        self.engines, self.stream, self.T = list(engines), e0.stream, len(engines)
        self.programs = {}          # (rows, smooth) -> [program handle, captured Graph or None]

    @staticmethod
    def shared_stream(device):
        return torch.cuda.Stream(device=device)

    def step(self, b, smooth=True):
        ops.tile_hint(self.engines[0].tile_mult)
        """One training step of every trial on its next ``b`` rows."""
        lib = _lib.load()
        key = (int(b), bool(smooth))
        with torch.cuda.stream(self.stream):
            plans = [e._pre_step(b, smooth) for e in self.engines]
            if key not in self.programs:
                # first step of this shape: every engine emits it eagerly (slab tables are recorded there), logged
                handles = (C.c_void_p * self.T)()
                counts = []
                for t, (e, P) in enumerate(zip(self.engines, plans)):
                    assert bool(smooth) not in P.graphs, "engine already stepped this shape on its own"
                    check(lib.raae_record_begin(), "raae_record_begin")
                    try:
                        e.emit_step(P, smooth, record=True)
                    finally:
                        h, n = C.c_void_p(), C.c_int(0)
                        rc = lib.raae_record_end(C.byref(h), C.byref(n))
                    check(rc, "raae_record_end (a launch of the step has no batched form)")
                    P.graphs[bool(smooth)] = None
                    handles[t] = h
                    counts.append(n.value)
                torch.cuda.synchronize(self.engines[0].device)
                prog = C.c_void_p()
                rc = lib.raae_multi_build(handles, self.T, C.byref(prog))
                for h in handles:
                    lib.raae_record_free(C.c_void_p(h))
                check(rc, f"raae_multi_build (the trials' steps differ: {counts} launches)")
                self.programs[key] = [prog, None]
                return
            prog, graph = self.programs[key]
            if graph is None:
                graph = ops.Graph()
                graph.begin()
                check(lib.raae_multi_launch(prog, C.c_void_p(self.stream.cuda_stream)), "raae_multi_launch")
                graph.end()
                self.programs[key][1] = graph
            graph.launch()
            for e in self.engines:
                e._count_bn_step(smooth)

    def validate(self, specs, auxs):
        """The per-epoch validation of every trial (``StepEngine.validate``: eval forwards, five losses, style metrics) as
        one launch sequence; ``specs`` / ``auxs``: each trial's resident validation split.  Returns ``[(z, losses)]``."""
        lib = _lib.load()
        ops.tile_hint(self.engines[0].tile_mult)
        key = ("val", int(specs[0].shape[0]), tuple(s.data_ptr() for s in specs))
        with torch.cuda.stream(self.stream):
            if key not in self.programs:
                handles, batched = [], True
                for t, e in enumerate(self.engines):
                    if batched:
                        check(lib.raae_record_begin(), "raae_record_begin")
                    try:
                        e.validate(specs[t], auxs[t], _phase="emit")
                    finally:
                        if batched:
                            h, n = C.c_void_p(), C.c_int(0)
                            rc = lib.raae_record_end(C.byref(h), C.byref(n))
                            if rc == 0:
                                handles.append(h)
                            else:
                                # a validation split of >= 1024 rows meets the conv networks' large-batch kernel
                                # instances, which have no batched form: the trials' validations then run one after the
                                # other on the shared stream (still one captured graph, and still on the device)
                                batched = False
                torch.cuda.synchronize(self.engines[0].device)
                prog = None
                if batched:
                    prog = C.c_void_p()
                    rc = lib.raae_multi_build((C.c_void_p * self.T)(*[h.value for h in handles]), self.T, C.byref(prog))
                for h in handles:
                    lib.raae_record_free(h)
                if batched:
                    check(rc, "raae_multi_build (validation)")
                self.programs[key] = [prog, None]
            else:
                prog, graph = self.programs[key]
                if graph is None:
                    graph = ops.Graph()
                    graph.begin()
                    if prog is not None:
                        check(lib.raae_multi_launch(prog, C.c_void_p(self.stream.cuda_stream)), "raae_multi_launch")
                    else:
                        for t, e in enumerate(self.engines):
                            e.validate(specs[t], auxs[t], _phase="emit")
                    graph.end()
                    self.programs[key][1] = graph
                graph.launch()
            return [e.validate(specs[t], auxs[t], _phase="read") for t, e in enumerate(self.engines)]

    def launches_per_step(self, b, smooth=True):
        prog = self.programs.get((int(b), bool(smooth)))
        return _lib.load().raae_multi_count(prog[0]) if prog else 0

    def release(self):
        lib = _lib.load()
        torch.cuda.synchronize(self.engines[0].device)
        for prog, graph in self.programs.values():
            del graph
            if prog is not None:
                lib.raae_multi_free(prog)
        self.programs = {}

    def __del__(self):
        try:
            self.release()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass
