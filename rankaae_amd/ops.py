"""Thin Python wrappers over the C ABI: torch tensors in (device memory + current stream
only -- PyTorch is plumbing here), HIP kernels underneath.  No fallback paths.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import BnT, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    assert t.is_cuda and t.is_contiguous(), "device, contiguous tensors only"
    if dtype is not None:
        assert t.dtype == dtype, (t.dtype, dtype)
    return C.c_void_p(t.data_ptr())


def make_bn(partials=None, nparts=0, count=0.0, running_mean=None, running_var=None, momentum=0.1, eps=1e-5,
            update_running=False):
    """Build a ``raae_bn_t``.  ``partials=None`` selects eval mode (running statistics)."""
    bn = BnT()
    bn.partials = partials.data_ptr() if partials is not None else None
    bn.nparts = int(nparts)
    bn.count = float(count)
    bn.running_mean = running_mean.data_ptr() if running_mean is not None else None
    bn.running_var = running_var.data_ptr() if running_var is not None else None
    bn.momentum = float(momentum)
    bn.eps = float(eps)
    bn.update_running = 1 if update_running else 0
    return bn


def _bnp(bn):
    return C.byref(bn) if bn is not None else None


def dense_fwd(x, B, K, in_kind, slope, bn, mask, w, bias, N, z, out_kind, out_slope=None, out_partials=None):
    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_dense_fwd(_ptr(x), B, K, in_kind, _ptr(slope), _bnp(bn), _ptr(mask), _ptr(w), _ptr(bias),
                                         N, _ptr(z), out_kind, _ptr(out_slope), _ptr(out_partials, torch.float64),
                                         C.byref(n), _stream()), "raae_dense_fwd")
        return n.value
    return _probed("dense_fwd_kernel", 4 * (B * K * (2 if mask is not None else 1) + N * K + N + B * N), launch)


def make_gen(gen):
    """``raae_maskgen_t`` from ``(state tensor [counter, seed, keys], slot offset, keep)`` or None (disabled): the
    step's hash keys live in the third 8-byte word of the engine's state; threshold and multiplier are formed here
    exactly as ``raae_rng_fill`` forms them from the float keep probability."""
    import numpy as np
    g = _lib.MaskGenT()
    if gen is not None:
        state, off, keep = gen
        assert state.dtype == torch.int64 and state.is_cuda and state.numel() >= 3 and 0 <= off < 2 ** 32 and 0.0 < keep <= 1.0
        k32 = np.float32(keep)
        g.keys, g.offset = state.data_ptr() + 16, int(off)
        g.thr = min(0xFFFFFFFF, int(float(k32) * 4294967296.0))
        g.inv = float(np.float32(1.0) / k32)
    return g


def dense_fwd_args(x, B, K, in_kind, slope, bn, mask, w, bias, N, z, out_kind, out_slope=None, out_partials=None,
                   storage=0, mask_scale=1.0, gen=None):
    """``raae_dense_fwd_t`` holding the arguments of ``dense_fwd`` (for ``dense_fwd_pair``).  ``storage``: RAAE_ST_*
    bits -- which of x / mask / z are bf16 tensors.  ``gen``: the layer's dropout multipliers are generated in the
    kernel (``make_gen``) instead of read from ``mask``; ``mask_scale``: what a bf16 {0, 1} mask is multiplied by."""
    a = _lib.DenseFwdT()
    a.storage = int(storage)
    a.mask_scale = float(mask_scale)
    a.gen = make_gen(gen)
    assert mask is None or gen is None
    a.x, a.B, a.K, a.in_kind, a.slope = _ptr(x, None), B, K, in_kind, _ptr(slope)
    a.has_bn = 0 if bn is None else 1
    if bn is not None:
        a.bn = bn
    a.mask, a.w, a.bias, a.N, a.z = _ptr(mask, None), _ptr(w), _ptr(bias), N, _ptr(z, None)
    a.out_kind, a.out_slope, a.out_partials = out_kind, _ptr(out_slope), _ptr(out_partials, torch.float64)
    for t, bit in ((x, _lib.ST_X), (mask, _lib.ST_MASK), (z, _lib.ST_Z)):      # the dtype of each tensor must match its bit
        assert t is None or t.dtype == (torch.bfloat16 if storage & bit else torch.float32), (t.dtype, storage, bit)
    return a


def dense_fwd_bytes(a):
    """Algorithmic bytes of a fused dense layer: input (and its dropout mask) read once, weights and bias once,
    output written once."""
    bx, bm, bz = (2 if a.storage & 1 else 4), (2 if a.storage & 2 else 4), (2 if a.storage & 4 else 4)
    # (dropout multipliers generated in the kernel are not bytes: only a mask TENSOR counts)
    return a.B * a.K * (bx + (bm if a.mask else 0)) + 4 * (a.N * a.K + a.N) + bz * a.B * a.N


def dense_fwd_struct(a):
    """``dense_fwd`` from a ``raae_dense_fwd_t``."""
    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_dense_fwd_s(C.byref(a), C.byref(n), _stream()), "raae_dense_fwd_s")
        return n.value
    return _probed("dense_fwd_kernel", dense_fwd_bytes(a), launch)


def dense_fwd_pair(p, q):
    """Two independent dense layers in one launch; returns both partial-row counts."""
    def launch():
        n1, n2 = C.c_int(0), C.c_int(0)
        check(_lib.load().raae_dense_fwd2(C.byref(p), C.byref(q), C.byref(n1), C.byref(n2), _stream()), "raae_dense_fwd2")
        return n1.value, n2.value
    return _probed("dense_fwd2_kernel", dense_fwd_bytes(p) + dense_fwd_bytes(q), launch)


def disc_fused(z_real, styles, noise, sigma, mask1, mask2, layers, alpha, n_real, n_fake, ns, gslab, slab_stride,
               dstyles, partial, ticket, loss):
    """The adversarial branch in one launch (``raae_disc_fused``); ``layers``: the three dense layers of the
    discriminator (``w``, ``b``, ``prelu``); returns the slab count of the parameter gradients."""
    l1, l2, l3 = layers
    a = _lib.DiscFusedT()
    a.z_real, a.styles, a.noise, a.sigma = z_real.data_ptr(), styles.data_ptr(), _p(noise), float(sigma)
    a.mask1, a.mask2 = _p(mask1), _p(mask2)
    a.w1, a.b1, a.s1 = l1.w.data_ptr(), l1.b.data_ptr(), l1.prelu.weight.data_ptr()
    a.w2, a.b2, a.s2 = l2.w.data_ptr(), l2.b.data_ptr(), l2.prelu.weight.data_ptr()
    a.w3, a.b3, a.alpha = l3.w.data_ptr(), l3.b.data_ptr(), alpha.data_ptr()
    a.n_real, a.n_fake, a.ns, a.hidden = n_real, n_fake, ns, l1.N
    a.dw1, a.db1, a.ds1 = gslab(l1.w).data_ptr(), gslab(l1.b).data_ptr(), gslab(l1.prelu.weight).data_ptr()
    a.dw2, a.db2, a.ds2 = gslab(l2.w).data_ptr(), gslab(l2.b).data_ptr(), gslab(l2.prelu.weight).data_ptr()
    a.dw3, a.db3 = gslab(l3.w).data_ptr(), gslab(l3.b).data_ptr()
    a.slab_stride = slab_stride
    a.dstyles, a.partial, a.ticket, a.loss = dstyles.data_ptr(), partial.data_ptr(), ticket.data_ptr(), loss.data_ptr()
    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_disc_fused(C.byref(a), C.byref(n), _stream()), "raae_disc_fused")
        return n.value
    rows = n_real + n_fake
    nmask = (1 if mask1 is not None else 0) + (1 if mask2 is not None else 0)
    # z_real / styles (+ noise) read, masks read, dstyles written, weights read + one gradient slab written per workgroup
    nbytes = 4 * (rows * ns * (2 if noise is not None else 1) + rows * l1.N * nmask + n_fake * ns +
                  2 * (l1.w.numel() + l2.w.numel() + l3.w.numel() + 3 * l1.N + 1))
    return _probed("disc_fused_kernel", nbytes, launch)


def dense_bwd(g, g_kind, g_partials, g_nparts, zout, out_slope, out_bn, B, N, x, K, in_kind, slope, bn, mask, w,
              dw, db, dslope, slab_stride, dx=None, dx_partials=None, storage=0, mask_scale=1.0, gen=None):
    for t, bit in ((x, _lib.ST_X), (mask, _lib.ST_MASK), (zout, _lib.ST_Z)):
        assert t is None or t.dtype == (torch.bfloat16 if storage & bit else torch.float32), (t.dtype, storage, bit)
    assert mask is None or gen is None
    a = _lib.DenseBwdT()
    a.g, a.g_kind, a.g_partials, a.g_nparts = _p(g), g_kind, _p(g_partials), int(g_nparts)
    a.zout, a.out_slope = _p(zout), _p(out_slope)
    a.has_out_bn = 0 if out_bn is None else 1
    if out_bn is not None:
        a.out_bn = out_bn
    a.B, a.N, a.x, a.K, a.in_kind, a.slope = B, N, _p(x), K, in_kind, _p(slope)
    a.has_bn = 0 if bn is None else 1
    if bn is not None:
        a.bn = bn
    a.mask, a.w, a.dw, a.db, a.dslope, a.slab_stride = _p(mask), _p(w), _p(dw), _p(db), _p(dslope), slab_stride
    a.dx, a.dx_partials, a.storage, a.mask_scale = _p(dx), _p(dx_partials), int(storage), float(mask_scale)
    a.gen = make_gen(gen)

    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_dense_bwd_s(C.byref(a), C.byref(n), _stream()), "raae_dense_bwd_s")
        return n.value
    # output gradient and raw output read, input (+ mask tensor) read, input gradient written, weights read, one slab written
    nbytes = 4 * (B * N * (2 if zout is not None else 1) + B * K * (2 if mask is not None else 1) +
                  (B * K if dx is not None else 0) + 2 * N * K + 2 * N)
    return _probed("dense_bwd_kernel", nbytes, launch)


def tail_prepare(src, dst, n, step_counter, rng_state, tail_state):
    check(_lib.load().raae_tail_prepare(_ptr(src), _ptr(dst), n, _ptr(step_counter, torch.int32), _ptr(rng_state, None),
                                        _ptr(tail_state, None), _stream()), "raae_tail_prepare")


def tile_hint(mult):
    """Thread-local launch-geometry hint of the conv-network entry points (``raae_tile_hint``)."""
    check(_lib.load().raae_tile_hint(int(mult)), "raae_tile_hint")


def stat_collapse2(p1, n1, C1, o1, p2=None, n2=0, C2=0, o2=None):
    def launch():
        check(_lib.load().raae_stat_collapse2(_ptr(p1, torch.float64), n1, C1, _ptr(o1, torch.float64),
                                              _ptr(p2, torch.float64), n2, C2, _ptr(o2, torch.float64), _stream()),
              "raae_stat_collapse2")
    _probed("stat_collapse_kernel", 16 * (n1 * C1 + n2 * C2), launch)


def style_bn_fwd(z, B, Cc, bn, styles):
    check(_lib.load().raae_style_bn_fwd(_ptr(z), B, Cc, _bnp(bn), _ptr(styles), _stream()), "raae_style_bn_fwd")


def style_bn_bwd(dstyles, styles, B, Cc, bn, dz, scale=1.0):
    check(_lib.load().raae_style_bn_bwd(_ptr(dstyles), _ptr(styles), B, Cc, _bnp(bn), scale, _ptr(dz), _stream()),
          "raae_style_bn_bwd")


def rank_loss_work_bytes(B, n_aux):
    return int(_lib.load().raae_rank_loss_work_bytes(B, n_aux))


def rank_loss_fwd_bwd(d, ldd, z, ldz, B, n_aux, activate, work, loss, dz):
    def launch():
        check(_lib.load().raae_rank_loss_fwd_bwd(_ptr(d), ldd, _ptr(z), ldz, B, n_aux, 1 if activate else 0,
                                                 _ptr(work, None), _ptr(loss), _ptr(dz), _stream()),
              "raae_rank_loss_fwd_bwd")
    _probed("rank_pairs_kernel", 4 * B * n_aux * 3, launch)      # O(B) bytes for B^2 n_aux pair operations: VALU-bound


def rank_rows_pairs(d_all, ldd, z_all, ldz, n_all, row0, nrows, n_aux, work, totals):
    check(_lib.load().raae_rank_rows_pairs(_ptr(d_all), ldd, _ptr(z_all), ldz, n_all, row0, nrows, n_aux, _ptr(work, None),
                                           _ptr(totals, torch.float64), _stream()), "raae_rank_rows_pairs")


def rank_rows_finish(totals, n_all, nrows, n_aux, activate, scale, work, loss, dz, ldz):
    check(_lib.load().raae_rank_rows_finish(_ptr(totals, torch.float64), n_all, nrows, n_aux, 1 if activate else 0,
                                            float(scale), _ptr(work, None), _ptr(loss), _ptr(dz), ldz, _stream()),
          "raae_rank_rows_finish")


def style_metrics(z, n, k, a_coef, work, out):
    check(_lib.load().raae_style_metrics(_ptr(z), n, k, _ptr(a_coef, torch.float64), _ptr(work, torch.float64),
                                         _ptr(out, torch.float64), _stream()), "raae_style_metrics")


def group_mean(x, groups, per, L, out):
    check(_lib.load().raae_group_mean(_ptr(x), groups, per, L, _ptr(out), _stream()), "raae_group_mean")


def _fin(fin):
    """``fin = (scale, out, slot, acc_slot, ticket)``: the loss is finished inside the kernel (``raae_loss_fin_t``);
    None: the partials are left for ``loss_finalize``."""
    if fin is None:
        return None
    scale, out, slot, acc_slot, ticket = fin
    f = _lib.LossFinT()
    f.scale, f.out, f.slot, f.acc_slot, f.ticket = float(scale), out.data_ptr(), slot, acc_slot, ticket.data_ptr()
    return C.byref(f)


def recon_loss_fwd_bwd(spec_in, spec_out, B, L, scale, partial, dout, fin=None):
    n = C.c_int(0)
    check(_lib.load().raae_recon_loss_fwd_bwd(_ptr(spec_in), _ptr(spec_out), B, L, 1 if scale else 0,
                                              _ptr(partial, torch.float64), C.byref(n), _ptr(dout), _fin(fin), _stream()),
          "raae_recon_loss_fwd_bwd")
    return n.value


def smooth_loss_fwd_bwd(x, B, L, taps, partial, dx, fin=None):
    n = C.c_int(0)
    arr = (C.c_float * len(taps))(*[float(t) for t in taps])
    check(_lib.load().raae_smooth_loss_fwd_bwd(_ptr(x), B, L, arr, len(taps), _ptr(partial, torch.float64),
                                               C.byref(n), _ptr(dx), _fin(fin), _stream()), "raae_smooth_loss_fwd_bwd")
    return n.value


def mse_fwd_bwd(a, b, n_el, partial, da, fin=None):
    n = C.c_int(0)
    check(_lib.load().raae_mse_fwd_bwd(_ptr(a), _ptr(b), n_el, _ptr(partial, torch.float64), C.byref(n), _ptr(da),
                                       _fin(fin), _stream()), "raae_mse_fwd_bwd")
    return n.value


def bce_pair_fwd_bwd(logits, n_real, n_fake, loss, dlogits):
    check(_lib.load().raae_bce_pair_fwd_bwd(_ptr(logits), n_real, n_fake, _ptr(loss), _ptr(dlogits), _stream()),
          "raae_bce_pair_fwd_bwd")


def disc_input(z_real, styles, noise, sigma, n_real, n_fake, Cc, out):
    check(_lib.load().raae_disc_input(_ptr(z_real), _ptr(styles), _ptr(noise), sigma, n_real, n_fake, Cc, _ptr(out),
                                      _stream()), "raae_disc_input")


def scale_by_dev(src, dev_scale, sign, n, dst):
    check(_lib.load().raae_scale_by_dev(_ptr(src), _ptr(dev_scale), sign, n, _ptr(dst), _stream()),
          "raae_scale_by_dev")


def loss_finalize(partial, n, scale, out, slot, acc_slot=-1):
    check(_lib.load().raae_loss_finalize(_ptr(partial, torch.float64), n, scale, _ptr(out), slot, acc_slot,
                                         _stream()), "raae_loss_finalize")


def gather_batch(spec, aux, idx, cursor, noise, spec_noise, B, L, n_aux, spec_out, aux_out):
    check(_lib.load().raae_gather_batch(_ptr(spec), _ptr(aux), _ptr(idx, torch.int64), _ptr(cursor, torch.int32),
                                        _ptr(noise), spec_noise, B, L, n_aux, _ptr(spec_out), _ptr(aux_out),
                                        _stream()), "raae_gather_batch")


def adam_step(p, m, v, g_slabs, slab_stride, seg_nslab, n, hyper, step, decoupled, max_nslab=512):
    def launch():
        check(_lib.load().raae_adam_step(_ptr(p), _ptr(m), _ptr(v), _ptr(g_slabs), slab_stride,
                                         _ptr(seg_nslab, torch.int16), n, _ptr(hyper, torch.float64),
                                         _ptr(step, torch.int32), 1 if decoupled else 0, int(max_nslab), _stream()),
              "raae_adam_step")
    _probed("adam_kernel", 28 * n, launch)       # p, m, v read + written, one gradient read (28 B / parameter)


def slab_reduce(g_slabs, slab_stride, seg_nslab, n, out, max_nslab=512):
    check(_lib.load().raae_slab_reduce(_ptr(g_slabs), slab_stride, _ptr(seg_nslab, torch.int16), n, _ptr(out),
                                       int(max_nslab), _stream()), "raae_slab_reduce")


def step_tick(steps, n, mask, rng_counter, cursor, cursor_inc):
    check(_lib.load().raae_step_tick(_ptr(steps, torch.int32), n, mask, _ptr(rng_counter, torch.int64),
                                     _ptr(cursor, torch.int32), cursor_inc, _stream()), "raae_step_tick")


def step_begin(steps, nsteps, mask, rng_state, cursor, stride, ticket, spec, aux, idx, B, L, n_aux, spec_noise, noise_tape,
               noise_goff, spec_out, aux_out, tape=None):
    """Tick + tape fill + batch gather in one launch (``raae_step_begin``).  ``noise_tape``: the noise slot of a
    host-filled tape (parity mode) or None (generated in the kernel at position ``noise_goff`` of the Gaussian
    numbering); ``tape``: an engine ``Tape`` whose resident slots are filled here (None / no slots: no fill)."""
    a = _lib.StepBeginT()
    a.steps, a.nsteps, a.step_mask, a.rng_state = _p(steps), int(nsteps), int(mask), _p(rng_state)
    a.cursor, a.stride, a.ticket = _p(cursor), int(stride), _p(ticket)
    a.spec, a.aux, a.idx, a.B, a.L, a.n_aux = _p(spec), _p(aux), _p(idx), B, L, n_aux
    a.spec_noise, a.noise_tape, a.noise_goff = float(spec_noise), _p(noise_tape), int(noise_goff)
    a.spec_out, a.aux_out = _p(spec_out), _p(aux_out)
    if tape is not None and len(tape.segs) > 0:
        a.tape, a.seg_desc, a.seg_scale, a.nseg, a.total = _p(tape.buf), _p(tape.seg_desc), _p(tape.seg_scale), len(tape.segs), tape.total
    assert steps.dtype == torch.int32 and rng_state.dtype == torch.int64 and cursor.dtype == torch.int32 and idx.dtype == torch.int64
    check(_lib.load().raae_step_begin(C.byref(a), _stream()), "raae_step_begin")


def rng_fill(tape, seg_desc, seg_scale, nseg, total, seed, counter):
    check(_lib.load().raae_rng_fill(_ptr(tape), _ptr(seg_desc, torch.int32), _ptr(seg_scale), nseg, total,
                                    C.c_ulonglong(seed), _ptr(counter, torch.int64), _stream()), "raae_rng_fill")


class Graph:
    """A captured HIP graph of one training step (hipStreamBeginCapture / hipGraphLaunch)."""

    active = 0      # captures in progress in this process (any engine, any thread): StepEngine.close() must not run inside one
    _lock = __import__("threading").Lock()

    def __init__(self):
        self.handle = C.c_void_p()

    def begin(self):
        check(_lib.load().raae_graph_begin(_stream()), "raae_graph_begin")
        with Graph._lock:
            Graph.active += 1

    def end(self):
        with Graph._lock:
            Graph.active -= 1
        check(_lib.load().raae_graph_end(_stream(), C.byref(self.handle)), "raae_graph_end")

    def launch(self):
        check(_lib.load().raae_graph_launch(self.handle, _stream()), "raae_graph_launch")

    def __del__(self):
        try:
            if self.handle:
                _lib.load().raae_graph_destroy(self.handle)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        check(_lib.load().raae_event_create(C.byref(self.h)), "raae_event_create")

    def record(self):
        check(_lib.load().raae_event_record(self.h, _stream()), "raae_event_record")

    def elapsed_ms(self, stop):
        ms = C.c_float(0)
        check(_lib.load().raae_event_elapsed_ms(self.h, stop.h, C.byref(ms)), "raae_event_elapsed_ms")
        return ms.value


# ------------------------------------------------------------------ roofline probe (bench.py)
class Probe:
    """While ``ops.PROBE`` holds one of these, every kernel launch that goes through a probed wrapper is followed
    by ``reps`` identical launches captured into a small hipGraph and replayed between two HIP events recorded on the
    stream the kernel is launched on (a single bracketed eager launch would measure the host's submission latency,
    tens of microseconds, not the kernel).  ``records[family] = [(event0, event1, algorithmic bytes)]``."""

    def __init__(self, reps=10, detail=False):
        self.reps, self.records, self.graphs, self.busy, self.detail = reps, {}, [], False, detail

    def summary(self):
        """``{family: dict(launches, avg_us, bytes)}`` -- call after a device synchronisation."""
        out = {}
        for fam, recs in self.records.items():
            us = [1e3 * a.elapsed_ms(z) / self.reps for a, z, _ in recs]
            out[fam] = {"launches": len(recs), "avg_us": sum(us) / len(us), "total_us": sum(us),
                        "bytes": sum(n for *_, n in recs) / len(recs)}
        return out


PROBE = None


def _probed(family, nbytes, launch, tag=None):
    out = launch()
    pr = PROBE
    if pr is None or pr.busy:
        return out
    if pr.detail and tag is not None:
        family = f"{family}[{tag}]"
    pr.busy = True
    try:
        g = Graph()
        g.begin()
        for _ in range(pr.reps):
            launch()
        g.end()
        e0, e1 = Event(), Event()
        # ALONE means alone: drain every stream first -- replays of a weight-gradient launch on a side stream were still
        # running beside the next main-stream kernel's replays (its "alone" time then doubled or not, depending on timing)
        torch.cuda.synchronize()
        e0.record()
        g.launch()
        e1.record()
        pr.graphs.append(g)
        pr.records.setdefault(family, []).append((e0, e1, int(nbytes)))
    finally:
        pr.busy = False
    return out


def _ktag(k):
    return f"{k.Cin}x{k.Lin}->{k.Cout}x{k.Lout}"


def block_bytes(kernel, k, B, mask=False, need_dx=True, input_bn=True):
    """ALGORITHMIC bytes of one launch of a fused residual-block kernel (SURVEY.md 8d's rule: every tensor the
    kernel must read or write counted once, fp32, weights once, no extra passes): ``k`` is the block's static
    shape (nets_conv.Block).  Per sample, with X [Cin,Lin], T1 [Cout,L1], everything else [C,Lout]:
      fwd_a  reads X (+ dropout mask)              writes T1, Sh (conv_short), E1, E2
      fwd_b  reads T1, E2, Sh | X                  writes T2, E3 (conv_excit), Y
      bwd_b  reads gY (+ Y behind a BatchNorm), T2, Sh, E3 | E2, T1 (, E2)      writes dT2, dSh, dEx, dBn2 (, dBnE)
      bwd_a  reads dBn2, T1, dSh, dBnE + E2 | dE2, E1 (, mask, X)               writes dT1 (, dE2), dE1, dR
    """
    ci_lin, co_l1, co_lo, ci_lo, ci_e = k.Cin * k.Lin, k.Cout * k.L1, k.Cout * k.Lout, k.Cin * k.Lout, k.Cin * k.E
    short, excit = k.cvs is not None, k.cve is not None
    nw = lambda mod: 0 if mod is None else mod.weight.numel() + (0 if mod.bias is None else mod.bias.numel())
    m = k.m
    if kernel == "fwd_a":
        per = ci_lin * (2 if mask else 1) + co_l1 + (co_lo if short else 0) + ci_e + ci_lo
        w = nw(m.conv1) + nw(m.conv_short) + nw(m.fc1) + nw(m.fc2)
    elif kernel == "fwd_b":
        per = co_l1 + ci_lo + (co_lo if short else ci_lin) + co_lo * (3 if excit else 2)
        w = nw(m.conv2) + nw(m.conv_excit)
    elif kernel == "bwd_b":
        per = co_lo * (2 if input_bn else 1) + co_lo * (3 if short else 2) + co_l1 + (ci_lo if excit else 0) + \
            3 * co_lo + co_l1 + (ci_lo if excit else 0)
        w = nw(m.conv2) + nw(m.conv_excit)
    elif kernel == "bwd_a":
        per = 2 * co_l1 + co_lo + (2 * ci_lo if excit else ci_lo) + ci_e + (ci_lin if mask else 0) + \
            co_l1 + (ci_lo if excit else 0) + ci_e + ((ci_lin * (2 if input_bn else 1)) if need_dx else 0)
        w = nw(m.conv1) + nw(m.conv_short) + nw(m.fc1) + nw(m.fc2)
    else:
        raise ValueError(kernel)
    return 4 * (B * per + w)


# ------------------------------------------------------------------ conv-network ops
def make_view(raw, slope=None, bn=None, mask=None):
    v = _lib.ViewT()
    v.raw = raw.data_ptr()
    v.slope = slope.data_ptr() if slope is not None else None
    if bn is not None:
        v.bn = bn
        v.has_bn = 1
    else:
        v.has_bn = 0
    v.mask = mask.data_ptr() if mask is not None else None
    v._keep = (raw, slope, mask)
    return v


def make_grad(g, raw=None, slope=None, bn=None, g_partials=None, g_nparts=0, u=None, act=0):
    s = _lib.GradT()
    s.g = g.data_ptr()
    s.g_partials = g_partials.data_ptr() if g_partials is not None else None
    s.g_nparts = int(g_nparts)
    s.u = u.data_ptr() if u is not None else None
    if bn is not None:
        s.bn = bn
        s.has_bn = 1
    else:
        s.has_bn = 0
    s.raw = raw.data_ptr() if raw is not None else None
    s.slope = slope.data_ptr() if slope is not None else None
    s.act = int(act)
    s._keep = (g, raw, slope, u, g_partials)
    return s


def make_conv(Cin, Lin, Cout, Lout, K, stride, pad, pad_replicate, groups, transposed):
    c = _lib.ConvT()
    c.Cin, c.Lin, c.Cout, c.Lout, c.K, c.stride, c.pad = Cin, Lin, Cout, Lout, K, stride, pad
    c.pad_replicate, c.groups, c.transposed = int(pad_replicate), groups, int(transposed)
    return c


def conv_bytes(cv, B, extra_in=0, extra_out=0):
    """``4 B (Cin Lin + Cout Lout) + 4 (weights + bias)`` per Conv1d call; ``extra_*``: further whole
    input- / output-sized tensors the call must touch (a mask, the raw output behind an activation gradient, ...)."""
    return 4 * B * (cv.Cin * cv.Lin * (1 + extra_in) + cv.Cout * cv.Lout * (1 + extra_out)) + \
        4 * (cv.Cout * (cv.Cin // cv.groups) * cv.K + cv.Cout)


def conv_fwd(view, B, cv, w, bias, out, stats_kind=0, out_slope=None, out_partials=None, act=0):
    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_conv_fwd(C.byref(view), B, C.byref(cv), _ptr(w), _ptr(bias), _ptr(out), stats_kind,
                                        _ptr(out_slope), _ptr(out_partials, torch.float64), C.byref(n), act, _stream()),
              "raae_conv_fwd")
        return n.value
    return _probed("conv_fwd (per-layer)", conv_bytes(cv, B, extra_in=1 if view.mask else 0), launch)


def conv_bwd_data(go, B, cv, w, view, din, accumulate, din_partials=None):
    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_conv_bwd_data(C.byref(go), B, C.byref(cv), _ptr(w), C.byref(view), _ptr(din),
                                             1 if accumulate else 0, _ptr(din_partials, torch.float64), C.byref(n),
                                             _stream()), "raae_conv_bwd_data")
        return n.value
    return _probed("conv_bwd_data (per-layer)", conv_bytes(cv, B, extra_in=1 if din_partials is not None else 0,
                                                           extra_out=1 if go.raw else 0), launch)


def head_bwd_supported(go, B, cv, view):
    return bool(_lib.load().raae_head_bwd_supported(C.byref(go), B, C.byref(cv), C.byref(view)))


def head_bwd(go, B, cv, w, view, din, din_partials, dw, dbias, slab_stride):
    """The decoder head's backward in one pass (``raae_head_bwd``): returns (partial rows of din, slabs of dw/dbias)."""
    def launch():
        n, ns = C.c_int(0), C.c_int(0)
        check(_lib.load().raae_head_bwd(C.byref(go), B, C.byref(cv), _ptr(w), C.byref(view), _ptr(din),
                                        _ptr(din_partials, torch.float64), C.byref(n), _ptr(dw), _ptr(dbias), slab_stride,
                                        C.byref(ns), _stream()), "raae_head_bwd")
        return n.value, ns.value
    # reads g, out and the C input rows, writes the C gradient rows
    return _probed("head_bwd", 4 * B * cv.Lin * (2 * cv.Cin + 2) + 4 * (cv.Cin + 1), launch)


def conv_bwd_weight(go, B, cv, view, dw, dbias, dslope, slab_stride):
    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_conv_bwd_weight(C.byref(go), B, C.byref(cv), C.byref(view), _ptr(dw), _ptr(dbias),
                                               _ptr(dslope), slab_stride, C.byref(n), _stream()), "raae_conv_bwd_weight")
        return n.value
    return _probed("conv_bwd_weight (per-layer)", conv_bytes(cv, B, extra_out=1 if go.raw else 0), launch)


def lenlin_fwd(view, B, Cc, Lin, w, bias, E, out, stats_kind=0, out_slope=None, out_partials=None):
    n = C.c_int(0)
    check(_lib.load().raae_lenlin_fwd(C.byref(view), B, Cc, Lin, _ptr(w), _ptr(bias), E, _ptr(out), stats_kind,
                                      _ptr(out_slope), _ptr(out_partials, torch.float64), C.byref(n), _stream()),
          "raae_lenlin_fwd")
    return n.value


def lenlin_bwd_data(go, B, Cc, E, w, view, Lin, din, accumulate, din_partials=None):
    n = C.c_int(0)
    check(_lib.load().raae_lenlin_bwd_data(C.byref(go), B, Cc, E, _ptr(w), C.byref(view), Lin, _ptr(din),
                                           1 if accumulate else 0, _ptr(din_partials, torch.float64), C.byref(n),
                                           _stream()), "raae_lenlin_bwd_data")
    return n.value


def lenlin_bwd_weight(go, B, Cc, E, view, Lin, dw, dbias, dslope, slab_stride):
    n = C.c_int(0)
    check(_lib.load().raae_lenlin_bwd_weight(C.byref(go), B, Cc, E, C.byref(view), Lin, _ptr(dw), _ptr(dbias),
                                             _ptr(dslope), slab_stride, C.byref(n), _stream()),
          "raae_lenlin_bwd_weight")
    return n.value


def sum3_fwd(va, vb, vc, B, Cc, L, y, out_partials=None):
    n = C.c_int(0)
    check(_lib.load().raae_sum3_fwd(C.byref(va), C.byref(vb), C.byref(vc), B, Cc, L, _ptr(y),
                                    _ptr(out_partials, torch.float64), C.byref(n), _stream()), "raae_sum3_fwd")
    return n.value


def grad_materialize(go, B, Cc, L, draw, accumulate=False, dslope=None, slab_stride=0):
    n = C.c_int(0)
    check(_lib.load().raae_grad_materialize(C.byref(go), B, Cc, L, _ptr(draw), 1 if accumulate else 0, _ptr(dslope),
                                            slab_stride, C.byref(n), _stream()), "raae_grad_materialize")
    return n.value


def _p(t):
    return t.data_ptr() if t is not None else None


def block_fwd_a_args(view_in, mask, B, k, m, T1, Sh, E1, E2, pT1, pE2):
    """``raae_block_fwd_a_t`` of a residual block (``k``: nets_conv.Block, ``m``: its nn.Module)."""
    a = _lib.BlockFwdAT()
    a.inp, a.mask, a.B = view_in, _p(mask), B
    a.Cin, a.Cout, a.Lin, a.L1, a.Lout, a.E = k.Cin, k.Cout, k.Lin, k.L1, k.Lout, k.E
    a.cv1 = k.cv1
    a.has_short = 1 if k.cvs is not None else 0
    if k.cvs is not None:
        a.cvs = k.cvs
        a.ws, a.bs = _p(m.conv_short.weight), _p(m.conv_short.bias)
    a.w1, a.b1, a.slope1 = _p(m.conv1.weight), _p(m.conv1.bias), _p(m.relu1.weight)
    a.wf1, a.bf1, a.se1 = _p(m.fc1.weight), _p(m.fc1.bias), _p(m.relu_excit_1.weight)
    a.wf2, a.bf2, a.se2 = _p(m.fc2.weight), _p(m.fc2.bias), _p(m.relu_excit_2.weight)
    a.T1, a.Sh, a.E1, a.E2, a.pT1, a.pE2 = _p(T1), _p(Sh), _p(E1), _p(E2), _p(pT1), _p(pE2)
    a.nbytes = block_bytes("fwd_a", k, B, mask=mask is not None)
    a.tag = _ktag(k)
    return a


def block_fwd_a(a):
    """Fused forward phase A of a residual block; returns the number of partial-statistic rows written."""
    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_block_fwd_a(C.byref(a), C.byref(n), _stream()), "raae_block_fwd_a")
        return n.value
    return _probed("block_fwd_a_kernel", a.nbytes, launch, a.tag)


def block_fwd_pair(kind, x, y):
    """Phase ``kind`` ("a" / "b") of two independent blocks in one launch; returns both row counts."""
    def launch():
        n1, n2 = C.c_int(0), C.c_int(0)
        fn = _lib.load().raae_block_fwd_a2 if kind == "a" else _lib.load().raae_block_fwd_b2
        check(fn(C.byref(x), C.byref(y), C.byref(n1), C.byref(n2), _stream()), f"raae_block_fwd_{kind}2")
        return n1.value, n2.value
    return _probed(f"block_fwd_{kind}2_kernel", x.nbytes + y.nbytes, launch)


def block_fwd_b_args(vT1, vE2, vR, B, k, m, Sh, T2, E3, Y, pY):
    a = _lib.BlockFwdBT()
    a.vT1, a.vE2 = vT1, vE2
    if vR is not None:
        a.vR = vR
    a.B, a.Cin, a.Cout, a.L1, a.Lout = B, k.Cin, k.Cout, k.L1, k.Lout
    a.cv2 = k.cv2
    a.has_short = 1 if k.cvs is not None else 0
    a.has_excit = 1 if k.cve is not None else 0
    if k.cve is not None:
        a.cve = k.cve
        a.we, a.be, a.se3 = _p(m.conv_excit.weight), _p(m.conv_excit.bias), _p(m.relu_excit_3.weight)
    if k.cvs is not None:
        a.Sh, a.ss = _p(Sh), _p(m.relu_short.weight)
    a.w2, a.b2, a.slope2 = _p(m.conv2.weight), _p(m.conv2.bias), _p(m.relu2.weight)
    a.T2, a.E3, a.Y, a.pY = _p(T2), _p(E3), _p(Y), _p(pY)
    a.nbytes = block_bytes("fwd_b", k, B)
    a.tag = _ktag(k)
    return a


def block_fwd_b(a):
    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_block_fwd_b(C.byref(a), C.byref(n), _stream()), "raae_block_fwd_b")
        return n.value
    return _probed("block_fwd_b_kernel", a.nbytes, launch, a.tag)


def block_bwd_b(gy, vT1, vE2, B, k, m, w, slab_stride, gslab, wgrad=None):
    """Fused backward phase B (``w``: the block's workspace, ``gslab(p)``: slab-0 view of parameter p)."""
    a = _lib.BlockBwdBT()
    a.gy, a.vT1 = gy, vT1
    if vE2 is not None:
        a.vE2 = vE2
    a.B, a.Cin, a.Cout, a.L1, a.Lout = B, k.Cin, k.Cout, k.L1, k.Lout
    a.cv2 = k.cv2
    a.has_short = 1 if k.cvs is not None else 0
    a.has_excit = 1 if k.cve is not None else 0
    a.w2, a.slope2, a.T2 = _p(m.conv2.weight), _p(m.relu2.weight), _p(w.T2)
    if k.cvs is not None:
        a.ss, a.Sh, a.dslope_s = _p(m.relu_short.weight), _p(w.Sh), _p(gslab(m.relu_short.weight))
    if k.cve is not None:
        a.cve = k.cve
        a.we, a.se, a.Ex = _p(m.conv_excit.weight), _p(m.relu_excit_3.weight), _p(w.E3)
        a.dslope_e = _p(gslab(m.relu_excit_3.weight))
        a.dBnE, a.pdBnE = _p(w.dBnE), _p(w.pdBnE)
    else:
        a.se, a.Ex = _p(m.relu_excit_2.weight), _p(w.E2)
        a.dslope_e = _p(gslab(m.relu_excit_2.weight))
    a.dT2, a.dSh, a.dEx, a.dBn2, a.pdBn2 = _p(w.dT2), _p(w.dSh), _p(w.dEx), _p(w.dBn2), _p(w.pdBn2)
    a.dslope2 = _p(gslab(m.relu2.weight))
    a.slab_stride = slab_stride
    nbytes = block_bytes("bwd_b", k, B, input_bn=bool(gy.has_bn))
    if wgrad is not None:          # the following block's weight-gradient tasks ride in the same launch
        def launch2():
            n, ns = C.c_int(0), (C.c_int * 6)()
            check(_lib.load().raae_block_bwd_b_wgrad(C.byref(a), C.byref(wgrad), C.byref(n), ns, _stream()),
                  "raae_block_bwd_b_wgrad")
            return n.value, list(ns)[:wgrad.n_conv + wgrad.n_lin]
        return _probed("block_bwd_b_wgrad_kernel", nbytes + wgrad.nbytes, launch2, _ktag(k))

    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_block_bwd_b(C.byref(a), C.byref(n), _stream()), "raae_block_bwd_b")
        return n.value
    return _probed("block_bwd_b_kernel", nbytes, launch, _ktag(k))


def block_bwd_a(g1, ge, view_in, mask, B, k, m, w, dE2, dR, pdR, slab_stride, gslab):
    a = _lib.BlockBwdAT()
    a.g1 = g1
    if ge is not None:
        a.ge = ge
    a.inp, a.mask = view_in, _p(mask)
    a.B, a.Cin, a.Cout, a.Lin, a.L1, a.Lout, a.E = B, k.Cin, k.Cout, k.Lin, k.L1, k.Lout, k.E
    a.cv1 = k.cv1
    a.has_short = 1 if k.cvs is not None else 0
    a.has_excit = 1 if k.cve is not None else 0
    if k.cvs is not None:
        a.cvs = k.cvs
        a.ws = _p(m.conv_short.weight)
    a.w1, a.wf1, a.wf2, a.se1 = _p(m.conv1.weight), _p(m.fc1.weight), _p(m.fc2.weight), _p(m.relu_excit_1.weight)
    a.E1, a.dSh, a.dT1, a.dE2, a.dE1, a.dR, a.pdR = _p(w.E1), _p(w.dSh), _p(w.dT1), _p(dE2), _p(w.dE1), _p(dR), _p(pdR)
    a.dslope1 = _p(gslab(m.relu1.weight))
    a.dslope_e1 = _p(gslab(m.relu_excit_1.weight))
    if k.cve is not None:
        a.dslope_e2 = _p(gslab(m.relu_excit_2.weight))
    a.slab_stride = slab_stride

    def launch():
        n = C.c_int(0)
        check(_lib.load().raae_block_bwd_a(C.byref(a), C.byref(n), _stream()), "raae_block_bwd_a")
        return n.value
    return _probed("block_bwd_a_kernel", block_bytes("bwd_a", k, B, mask=mask is not None, need_dx=dR is not None,
                                                     input_bn=pdR is not None), launch, _ktag(k))


def block_wgrad_args(B, conv_tasks, lin_tasks, slab_stride):
    """The ``raae_block_wgrad_t`` of ``block_wgrad`` (to be launched later, e.g. inside ``block_bwd_b(wgrad=...)``)."""
    a = _lib.BlockWgradT()
    a.n_conv, a.n_lin, a.B, a.slab_stride = len(conv_tasks), len(lin_tasks), B, slab_stride
    for i, (go, cv, view, dw, db) in enumerate(conv_tasks):
        a.conv[i].go, a.conv[i].cv, a.conv[i].inp = go, cv, view
        a.conv[i].dw, a.conv[i].dbias = dw.data_ptr(), db.data_ptr()
    for i, (go, Cc, E, Lin, view, dw, db) in enumerate(lin_tasks):
        a.lin[i].go, a.lin[i].C, a.lin[i].E, a.lin[i].Lin, a.lin[i].inp = go, Cc, E, Lin, view
        a.lin[i].dw, a.lin[i].dbias = dw.data_ptr(), db.data_ptr()
    # algorithmic bytes: every DISTINCT tensor the launch must read counted once (the block input is the
    # operand of up to three tasks -- conv1, conv_short, fc1 -- and is one tensor; VERDICT r2: 74 MB for the last
    # decoder block at 4096 rows, not the 101 MB of the per-task sum) + one slab of weight and bias gradients per task
    uniq = {}
    for go, cv, view, _, _ in conv_tasks:
        uniq[go.g] = 4 * B * cv.Cout * cv.Lout
        uniq[view.raw] = 4 * B * cv.Cin * cv.Lin
        if view.mask:
            uniq[view.mask] = 4 * B * cv.Cin * cv.Lin
    for go, Cc, E, Lin, view, _, _ in lin_tasks:
        uniq[go.g] = 4 * B * Cc * E
        uniq[view.raw] = 4 * B * Cc * Lin
        if view.mask:
            uniq[view.mask] = 4 * B * Cc * Lin
    a.nbytes = sum(uniq.values()) + \
        sum(4 * (cv.Cout * (cv.Cin // cv.groups) * cv.K + cv.Cout) for _, cv, _, _, _ in conv_tasks) + \
        sum(4 * (E * Lin + E) for _, Cc, E, Lin, _, _, _ in lin_tasks)
    if conv_tasks:
        cv = conv_tasks[0][1]
        a.tag = f"{len(conv_tasks)}conv+{len(lin_tasks)}lin {cv.Cin}x{cv.Lin}->{cv.Cout}x{cv.Lout}"
    return a


def block_wgrad(B, conv_tasks, lin_tasks, slab_stride, args=None):
    """``conv_tasks``: [(grad, conv_desc, view, dw, dbias)], ``lin_tasks``: [(grad, C, E, Lin, view, dw, dbias)].
    One launch; returns the slab count per task (conv tasks first)."""
    a = args if args is not None else block_wgrad_args(B, conv_tasks, lin_tasks, slab_stride)

    def launch():
        ns = (C.c_int * 6)()
        check(_lib.load().raae_block_wgrad(C.byref(a), ns, _stream()), "raae_block_wgrad")
        return list(ns)[:a.n_conv + a.n_lin]
    return _probed("wgrad_multi_kernel", a.nbytes, launch, getattr(a, "tag", None))
