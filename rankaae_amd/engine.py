"""The MI355X step engine: the five-phase adversarial-autoencoder training step of the
reference (``sc/clustering/trainer.py:103-204``) as an explicit forward/backward program
of fused HIP kernels over a flat parameter arena, captured once per batch shape into a
hipGraph and replayed.

Design (DESIGN.md has the long form):
  * parameters live in ONE fp32 arena ordered [discriminator | encoder | decoder] so every
    optimizer of the reference (trainer.py:333-397) covers one contiguous range; the
    ``nn.Module`` containers hold views into it;
  * parameter gradients leave each backward kernel as per-workgroup *slabs* that the fused
    Adam kernel sums in fixed order -- no atomics, no zeroing, bitwise reproducible;
  * every random number of a step (spectral noise, dropout scales, latent samples) is a
    slot of one *tape* buffer: ``rng_mode="philox"`` fills it on the device inside the
    graph; ``rng_mode="host"`` draws it from the global torch CPU generator in the
    reference's consumption order (parity mode) and uploads it;
  * per-step scalars that change between replays (Adam step counts, learning rates, the
    gradient-reversal alpha, the epoch row cursor) live in device memory.
PyTorch is used for device memory and the current stream only.
"""
import math

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from .metrics import StyleMetrics
from ._lib import (IN_NONE, IN_PRELU_BN_DROP, IN_PRELU_DROP, OUT_RAW, OUT_STATS_PRELU, OUT_STATS_RAW, OUT_SOFTPLUS,
                   OUT_RELU, G_DIRECT, G_SOFTPLUS, G_PRELU_BN, G_PRELU, G_RELU, RAAE_MAX_PARTS)

PROBE_REPS = 10
OPT_NAMES = ["adversarial", "correlation", "reconstruction", "mutual_info", "smoothness"]
LOSS_SLOTS = {"adversarial": 0, "kendall": 1, "recon": 2, "mutual_info": 3, "smooth": 4, "mi_accum": 5}


def gaussian_taps(kernel_size=17, sigma=3.0):
    """Normalised Gaussian taps in float32 (what ``GaussianSmoothing`` builds, reference
    ``model.py:187-200``): computed here with numpy float32 arithmetic."""
    t = np.arange(kernel_size, dtype=np.float32)
    mean = np.float32((kernel_size - 1) / 2)
    k = np.float32(1 / (sigma * math.sqrt(2 * math.pi))) * np.exp(
        -(((t - mean) / np.float32(sigma)) ** 2) / np.float32(2)).astype(np.float32)
    return (k / k.sum(dtype=np.float32)).astype(np.float32)


# ------------------------------------------------------------------------------ arena
class Arena:
    """Flat fp32 parameter arena; tensors start at multiples of 64 floats."""

    def __init__(self, named_modules, device):
        self.device = device
        self.ranges, self.offsets, plan, off = {}, {}, [], 0
        for name, mod in named_modules:
            lo = off
            for p in mod.parameters():
                plan.append((p, off))
                self.offsets[id(p)] = off
                off += (p.numel() + 63) // 64 * 64
            self.ranges[name] = (lo, off)
        self.n = off
        self.P = torch.zeros(self.n, dtype=torch.float32, device=device)
        for p, o in plan:
            view = self.P[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
        for _, mod in named_modules:      # buffers (BN running stats) just move to the device
            for b in mod.buffers():
                b.data = b.data.to(device)

    def off(self, p):
        return self.offsets[id(p)]


class OptState:
    """One optimizer of ``Trainer.load_optimizers``: a contiguous arena range + moments."""

    def __init__(self, index, name, lo, hi, lr, betas, eps, wd, device):
        self.index, self.name, self.lo, self.hi = index, name, lo, hi
        self.lr, self.betas, self.eps, self.wd = float(lr), betas, float(eps), float(wd)
        self.m = torch.zeros(hi - lo, device=device)
        self.v = torch.zeros(hi - lo, device=device)
        self.hyper = torch.zeros(5, dtype=torch.float64, device=device)
        self.push()

    def push(self):
        self.hyper.copy_(torch.tensor([self.lr, self.betas[0], self.betas[1], self.eps, self.wd],
                                      dtype=torch.float64))


# ------------------------------------------------------------------------------ tape
class Tape:
    """Per-step random tape: named slots (4-float aligned) + the reference's draw order."""

    def __init__(self):
        self.total = 0
        self.htotal = 0     # dropout elements numbered so far (resident and virtual slots alike)
        self.gtotal = 0     # Gaussian elements numbered so far
        self.hash_off = {}
        self.segs = []      # (offset, count, kind, keep, hash offset)   kind 0 = N(0,1), 1 = dropout scale
        self.draws = []     # in reference order: ("normal"|"mask", offset, shape, keep) | ("int64",)
        self.buf = None

    def slot(self, count, kind, keep=1.0, virtual=False):
        """``kind`` 0: N(0,1) floats; 1: dropout multipliers (0 or 1/keep) as floats; 2: dropout keep flags {0, 1} stored
        as bf16 (``precision: bf16``; the consumer multiplies by the fp32 1/keep) -- ``count`` values in
        ``(count + 1) // 2`` floats of the buffer.  Returns the slot's tape offset; ``self.hash_off[offset]`` is the
        slot's position in the numbering of all dropout elements of a step (what the hash that decides an element is
        keyed with).  ``virtual``: a kind-1 slot whose multipliers the consumer generates itself (``raae_maskgen_t``):
        no tape memory, nothing to fill; returns ``-(hash offset) - 1``."""
        if kind == 0:       # Gaussian slots are numbered apart (their Philox counter = numbering / 4), so that the noise
            hoff = self.gtotal          # does not depend on which dropout slots live on the tape
            self.gtotal += (count + 3) // 4 * 4
        else:
            hoff = self.htotal
            self.htotal += (count + 3) // 4 * 4
            assert self.htotal < (1 << 31)
        if virtual:         # kind 1: generated by the consuming kernel; kind 0: the spectral noise, generated by raae_step_begin
            assert kind in (0, 1)
            return -hoff - 1
        off = self.total
        nfloat = (count + 1) // 2 if kind == 2 else count
        self.segs.append((off, nfloat, kind, keep, hoff))
        self.total += (nfloat + 3) // 4 * 4
        self.hash_off[off] = hoff
        return off

    def draw(self, kind, off, shape, keep=1.0):
        self.draws.append((kind, off, tuple(shape), keep))

    def finalize(self, device):
        self.buf = torch.zeros(max(self.total, 4), device=device)
        d = torch.tensor([[o, c, k, h] for o, c, k, _, h in self.segs], dtype=torch.int32)
        self.seg_desc = d.to(device)
        self.seg_scale = torch.tensor([kp for _, _, _, kp, _ in self.segs], dtype=torch.float32).to(device)
        self.host = torch.zeros(max(self.total, 4)).pin_memory() if torch.cuda.is_available() else None

    def view(self, off, *shape):
        n = int(np.prod(shape))
        return self.buf[off:off + n].view(*shape)

    def view16(self, off, *shape):
        """bf16 view of a kind-2 slot."""
        n = int(np.prod(shape))
        return self.buf[off:off + (n + 1) // 2].view(torch.bfloat16)[:n].view(*shape)

    def fill_host(self):
        """Parity mode: draw from the GLOBAL torch CPU generator exactly as the reference's
        ``randn_like`` / ``nn.Dropout`` / ``randn`` calls would (finding 9)."""
        for kind, off, shape, keep in self.draws:
            n = int(np.prod(shape))
            if kind == "normal":
                self.host[off:off + n] = torch.randn(*shape).reshape(-1)
            elif kind == "mask":
                self.host[off:off + n] = torch.empty(*shape).bernoulli_(keep).div_(keep).reshape(-1)
            elif kind == "mask16":      # the same draw, stored as bf16 keep flags {0, 1} (the kernels multiply by 1/keep)
                m16 = torch.empty(*shape).bernoulli_(keep).reshape(-1).to(torch.bfloat16)
                self.host[off:off + (n + 1) // 2].view(torch.bfloat16)[:n] = m16
            else:
                raise ValueError(kind)
        self.buf.copy_(self.host, non_blocking=True)


# ------------------------------------------------------------------------------ dense nets
class DenseLayer:
    def __init__(self, lin):
        self.lin, self.w, self.b = lin, lin.weight, lin.bias
        self.N, self.K = lin.out_features, lin.in_features
        self.prelu = None      # nn.PReLU following this layer
        self.bn = None         # nn.BatchNorm1d following the PReLU (or the Linear, for the style layer)
        self.p = 0.0           # dropout after it


def dense_layers(seq):
    layers = []
    for m in seq:
        if isinstance(m, nn.Linear):
            layers.append(DenseLayer(m))
        elif isinstance(m, nn.PReLU):
            layers[-1].prelu = m
        elif isinstance(m, nn.BatchNorm1d):
            layers[-1].bn = m
        elif isinstance(m, nn.Dropout):
            layers[-1].p = m.p
    return layers


class FCNet:
    """Emitter for ``FCEncoder`` / ``FCDecoder``: one fused dense kernel per layer."""

    def __init__(self, module, kind, eng):
        self.module, self.kind, self.eng = module, kind, eng
        self.layers = dense_layers(module.main)
        self.in_dim, self.out_dim = self.layers[0].K, self.layers[-1].N
        self.final_relu = kind == "dec" and isinstance(module.main[-1], nn.ReLU)
        self.bn_modules = [l.bn for l in self.layers if l.bn is not None]
        self.pairable = True       # forward_steps yields at every layer, the first time before anything is written
        # `precision: bf16`: hidden activations and dropout multipliers are STORED as bf16 (all arithmetic fp32)
        self.bf16 = bool(getattr(eng, "bf16", False))

    def alloc(self, b):
        dev = self.eng.device
        ws = type("WS", (), {})()
        ws.b = b
        hid = torch.bfloat16 if self.bf16 else torch.float32
        # (first and last layer outputs stay fp32: the first layer sees the un-normalised input -- spectra that differ
        # by a fraction of a percent -- so its pre-activations vary by less than a bf16 step across the batch and the
        # BatchNorm behind them would amplify the rounding to O(1): measured, 40-70 % gradient error; every deeper
        # layer reads BatchNorm-normalised inputs.  The last layer feeds the losses.)
        ws.z = [torch.empty(b, l.N, device=dev, dtype=hid if 0 < i < len(self.layers) - 1 else torch.float32)
                for i, l in enumerate(self.layers)]
        ws.part = [torch.zeros(RAAE_MAX_PARTS, l.N, 2, dtype=torch.float64, device=dev) for l in self.layers]
        ws.nparts = [0] * len(self.layers)
        wmax = max(max(l.N, l.K) for l in self.layers)
        ws.dx = [torch.empty(b, wmax, device=dev) for _ in range(2)]
        ws.dxp = [torch.zeros(RAAE_MAX_PARTS, wmax, 2, dtype=torch.float64, device=dev) for _ in range(2)]
        if self.kind == "enc":
            ws.styles = torch.empty(b, self.out_dim, device=dev)
            ws.dz_last = torch.empty(b, self.out_dim, device=dev)
            ws.out = ws.styles
        else:
            ws.out = ws.z[-1]
        return ws

    def mask_slots(self, tape, b, train=True):
        """Allocate + record (in forward order) the dropout masks of one forward pass."""
        masks = []
        inline = self.eng.inline_masks and not self.bf16
        for l in self.layers[:-1]:
            if train and l.p > 0:
                off = tape.slot(b * l.N, 2 if self.bf16 else 1, 1.0 - l.p, virtual=inline)
                if not inline:
                    tape.draw("mask16" if self.bf16 else "mask", off, (b, l.N), 1.0 - l.p)
                masks.append((off, (b, l.N), 1.0 - l.p, inline))
            else:
                masks.append(None)
        return masks

    def _mask(self, masks, i):
        """``dict(mask=, gen=, mask_scale=)`` of layer i's dropout multipliers: a tape view, or the in-kernel generator."""
        off, shape, keep, inline = masks[i]
        if inline:
            return dict(mask=None, gen=(self.eng.gen_state, -off - 1, keep), mask_scale=1.0)
        if self.bf16:
            return dict(mask=self.eng.tape.view16(off, *shape), gen=None, mask_scale=1.0 / keep)
        return dict(mask=self.eng.tape.view(off, *shape), gen=None, mask_scale=1.0)

    def _bn_in(self, ws, i, train, update):
        p = self.layers[i]
        if train:
            return ops.make_bn(ws.part[i], ws.nparts[i], ws.b, p.bn.running_mean, p.bn.running_var,
                               p.bn.momentum, p.bn.eps, update)
        return ops.make_bn(None, 0, 0, p.bn.running_mean, p.bn.running_var, p.bn.momentum, p.bn.eps, False)

    def forward(self, ws, x, masks, train=True):
        """One forward pass, every layer its own launch."""
        steps = self.forward_steps(ws, x, masks, train)
        try:
            _, args, nbytes = next(steps)
            while True:
                _, args, nbytes = steps.send(ops.dense_fwd_struct(args))
        except StopIteration as done:
            return done.value

    @staticmethod
    def forward_pair(first, second):
        """Two independent forward passes (``forward_steps`` generators) in lockstep: layer i of both in one launch
        (raae_dense_fwd2) while both have layers left.  Returns the two outputs."""
        gens, cur, out = [first, second], [None, None], [None, None]

        def advance(j, n):
            try:
                cur[j] = next(gens[j]) if n is None else gens[j].send(n)
            except StopIteration as done:
                out[j], gens[j], cur[j] = done.value, None, None
        advance(0, None)
        advance(1, None)
        while gens[0] is not None or gens[1] is not None:
            if gens[0] is not None and gens[1] is not None:
                n1, n2 = ops.dense_fwd_pair(cur[0][1], cur[1][1])
                advance(0, n1)
                advance(1, n2)
            else:
                j = 0 if gens[0] is not None else 1
                advance(j, ops.dense_fwd_struct(cur[j][1]))
        return out[0], out[1]

    def forward_steps(self, ws, x, masks, train=True):
        """Generator form of the forward pass: yields ``("dense", args, algorithmic bytes)`` per layer and expects
        the launch's partial-row count back."""
        eng, L, b = self.eng, self.layers, ws.b
        for i, l in enumerate(L):
            last = i == len(L) - 1
            if i == 0:
                xin, in_kind, slope, bn, mask = x, IN_NONE, None, None, None
            else:
                p = L[i - 1]
                xin, in_kind, slope = ws.z[i - 1], IN_PRELU_BN_DROP, p.prelu.weight
                bn = self._bn_in(ws, i - 1, train, True)
                mk = self._mask(masks, i - 1) if (train and masks[i - 1]) else None
                mask = mk["mask"] if mk else None
            if not last:
                out_kind, oslope = OUT_STATS_PRELU, l.prelu.weight
            elif self.kind == "enc":
                out_kind, oslope = OUT_STATS_RAW, None
            else:
                out_kind, oslope = (OUT_RELU if self.final_relu else OUT_SOFTPLUS), None
            st = 0
            if self.bf16:
                st = (_lib.ST_X if i > 1 else 0) | (_lib.ST_MASK if mask is not None else 0) | (_lib.ST_Z if 0 < i and not last else 0)
            extra = dict(gen=mk["gen"], mask_scale=mk["mask_scale"]) if (i > 0 and mk) else {}
            ws.nparts[i] = yield ("dense", ops.dense_fwd_args(xin, b, l.K, in_kind, slope, bn, mask, l.w, l.b, l.N,
                                                               ws.z[i], out_kind, oslope, ws.part[i], storage=st, **extra), 0)
        if self.kind == "enc":
            ops.style_bn_fwd(ws.z[-1], b, self.out_dim, self._bn_in(ws, len(L) - 1, train, True), ws.styles)
        if train:
            eng.count_bn(self.bn_modules)
        return ws.out

    def backward(self, ws, x, masks, g_out, dx_in=None, pending=None, keep_pending=False):
        """``g_out``: dL/d(output).  Writes parameter-gradient slabs; returns dL/d(input) in
        ``dx_in`` (if given).  Records the slab count of every parameter it touched."""
        assert pending is None     # (the conv networks hand weight-gradient tasks across; the dense ones have none)
        eng, L, b = self.eng, self.layers, ws.b
        n = len(L)
        if self.kind == "enc":
            ops.style_bn_bwd(g_out, ws.styles, b, self.out_dim, self._bn_in(ws, n - 1, True, False), ws.dz_last)
            g, gk, gp, gnp = ws.dz_last, G_DIRECT, None, 0
        else:
            g, gk, gp, gnp = g_out, (G_RELU if self.final_relu else G_SOFTPLUS), None, 0
        for i in reversed(range(n)):
            l = L[i]
            out_bn = self._bn_in(ws, i, True, False) if gk == G_PRELU_BN else None
            oslope = l.prelu.weight if gk == G_PRELU_BN else None
            if i == 0:
                xin, in_kind, slope, bn, mask, dx, dxp = x, IN_NONE, None, None, None, dx_in, None
            else:
                p = L[i - 1]
                xin, in_kind, slope = ws.z[i - 1], IN_PRELU_BN_DROP, p.prelu.weight
                bn = self._bn_in(ws, i - 1, True, False)
                mk = self._mask(masks, i - 1) if masks[i - 1] else None
                mask = mk["mask"] if mk else None
                dx, dxp = ws.dx[i & 1], ws.dxp[i & 1]
            ds = eng.gslab(l.prelu.weight) if gk == G_PRELU_BN else None
            st = 0
            if self.bf16:
                st = (_lib.ST_X if i > 1 else 0) | (_lib.ST_MASK if mask is not None else 0) | (_lib.ST_Z if 0 < i < n - 1 else 0)
            extra = dict(gen=mk["gen"], mask_scale=mk["mask_scale"]) if (i > 0 and mk) else {}
            ns = ops.dense_bwd(g, gk, gp, gnp, ws.z[i], oslope, out_bn, b, l.N, xin, l.K, in_kind, slope, bn, mask,
                               l.w, eng.gslab(l.w), eng.gslab(l.b), ds, eng.arena.n, dx, dxp, storage=st, **extra)
            eng.note_slabs([l.w, l.b] + ([l.prelu.weight] if ds is not None else []), ns)
            g, gk, gp, gnp = dx, G_PRELU_BN, dxp, ns


class DiscNet:
    """Emitter for ``DiscriminatorFC`` on the concatenated [real; fake] batch."""

    def __init__(self, module, eng):
        self.module, self.eng = module, eng
        self.layers = dense_layers(module.main)
        self.nstyle = self.layers[0].K

    def alloc(self, n_real, n_fake):
        dev = self.eng.device
        ws = type("WS", (), {})()
        ws.n_real, ws.n_fake, ws.n = n_real, n_fake, n_real + n_fake
        ws.x = torch.empty(ws.n, self.nstyle, device=dev)
        ws.z = [torch.empty(ws.n, l.N, device=dev) for l in self.layers]
        ws.dlogit = torch.empty(ws.n, 1, device=dev)
        wmax = max(max(l.N, l.K) for l in self.layers)
        ws.dx = [torch.empty(ws.n, wmax, device=dev) for _ in range(2)]
        ws.dstyles = torch.empty(n_fake, self.nstyle, device=dev)
        ws.partial = torch.zeros(256, dtype=torch.float64, device=dev)       # raae_disc_fused: loss partials
        ws.ticket = torch.zeros(1, dtype=torch.int32, device=dev)           # ... and its arrival counter
        return ws

    def tape_slots(self, tape, n_real, n_fake, train):
        """Slots for z_real, input noise and dropout masks; draws recorded in the reference's
        order: z_real; D(z_real): noise, masks; D(styles): noise, masks (functions.py:119-127)."""
        n, ns = n_real + n_fake, self.nstyle
        sl = type("S", (), {})()
        sl.z_real = tape.slot(n_real * ns, 0)
        tape.draw("normal", sl.z_real, (n_real, ns))
        sl.noise, sl.masks = None, [None] * (len(self.layers) - 1)
        if train:
            sl.noise = tape.slot(n * ns, 0)
            for i, l in enumerate(self.layers[:-1]):
                if l.p > 0:
                    sl.masks[i] = (tape.slot(n * l.N, 1, 1.0 - l.p), (n, l.N))
            for rows, r0 in ((n_real, 0), (n_fake, n_real)):
                tape.draw("normal", sl.noise + r0 * ns, (rows, ns))
                for i, l in enumerate(self.layers[:-1]):
                    if l.p > 0:
                        tape.draw("mask", sl.masks[i][0] + r0 * l.N, (rows, l.N), 1.0 - l.p)
        return sl

    def forward_backward(self, ws, sl, styles, loss_out, train=True):
        eng, L = self.eng, self.layers
        tape = eng.tape
        z_real = tape.view(sl.z_real, ws.n_real, self.nstyle)
        noise = tape.view(sl.noise, ws.n, self.nstyle) if train else None
        if (train and len(L) == 3 and L[0].N == 64 and L[1].N == 64 and L[2].N == 1 and self.nstyle <= 16 and
                eng.cfg.get("fused_discriminator", True)):
            # the reference's discriminator shape: the whole branch (input, 3 layers forward, BCE, backward, gradient
            # reversal) is one launch
            masks = [tape.view(m[0], *m[1]) if m else None for m in sl.masks]
            ns_ = ops.disc_fused(z_real, styles, noise, float(self.module.noise), masks[0], masks[1], L, eng.alpha_dev,
                                 ws.n_real, ws.n_fake, self.nstyle, eng.gslab, eng.arena.n, ws.dstyles, ws.partial,
                                 ws.ticket, loss_out)
            eng.note_slabs([L[0].w, L[0].b, L[0].prelu.weight, L[1].w, L[1].b, L[1].prelu.weight, L[2].w, L[2].b], ns_)
            return ws.dstyles
        ops.disc_input(z_real, styles, noise, float(self.module.noise), ws.n_real, ws.n_fake, self.nstyle, ws.x)
        masks = [tape.view(m[0], *m[1]) if (train and m) else None for m in sl.masks]
        for i, l in enumerate(L):
            if i == 0:
                ops.dense_fwd(ws.x, ws.n, l.K, IN_NONE, None, None, None, l.w, l.b, l.N, ws.z[0], OUT_RAW)
            else:
                ops.dense_fwd(ws.z[i - 1], ws.n, l.K, IN_PRELU_DROP, L[i - 1].prelu.weight, None, masks[i - 1], l.w,
                              l.b, l.N, ws.z[i], OUT_RAW)
        ops.bce_pair_fwd_bwd(ws.z[-1], ws.n_real, ws.n_fake, loss_out, ws.dlogit if train else None)
        if not train:
            return None
        g, gk = ws.dlogit, G_DIRECT
        for i in reversed(range(len(L))):
            l = L[i]
            oslope = l.prelu.weight if gk == G_PRELU else None
            ds = eng.gslab(l.prelu.weight) if gk == G_PRELU else None
            dx = ws.dx[i & 1]
            if i == 0:
                ns = ops.dense_bwd(g, gk, None, 0, ws.z[0], oslope, None, ws.n, l.N, ws.x, l.K, IN_NONE, None, None,
                                   None, l.w, eng.gslab(l.w), eng.gslab(l.b), ds, eng.arena.n, dx, None)
            else:
                ns = ops.dense_bwd(g, gk, None, 0, ws.z[i], oslope, None, ws.n, l.N, ws.z[i - 1], l.K, IN_PRELU_DROP,
                                   L[i - 1].prelu.weight, None, masks[i - 1], l.w, eng.gslab(l.w), eng.gslab(l.b), ds,
                                   eng.arena.n, dx, None)
            eng.note_slabs([l.w, l.b] + ([l.prelu.weight] if ds is not None else []), ns)
            g, gk = dx, G_PRELU
        # gradient reversal: d styles = -alpha * dL/dx of the fake rows (model.py:15-22)
        fake = g.view(-1)[ws.n_real * self.nstyle:(ws.n_real + ws.n_fake) * self.nstyle]
        ops.scale_by_dev(fake, eng.alpha_dev, -1.0, ws.n_fake * self.nstyle, ws.dstyles)
        return ws.dstyles


# ------------------------------------------------------------------------------ the engine
def _on_stream(fn):
    """Run a StepEngine method on the engine's private HIP stream."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        ops.tile_hint(self.tile_mult)        # thread-local in the library: whatever engine ran last on this thread set its own
        with torch.cuda.stream(self.stream):
            if fn.__name__ != "step" and getattr(self, "_tail", None) is not None:
                self._run_tail()
            return fn(self, *a, **kw)
    return wrapper


def _on_stream_io(fn):
    """Like ``_on_stream`` for methods that take and return caller tensors: the engine's stream first waits for
    the caller's current stream (inputs written there) and the caller's stream then waits for the engine's
    (results read there) -- stream-ordered, no host synchronisation."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        ops.tile_hint(self.tile_mult)
        caller = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            if getattr(self, "_tail", None) is not None:
                self._run_tail()
            out = fn(self, *a, **kw)
        caller.wait_stream(self.stream)
        return out
    return wrapper


class StepPlan:
    """Everything that depends on the batch size: workspaces, tape slots, captured graphs."""
    pass


class StepEngine:
    def __init__(self, encoder, decoder, discriminator, cfg, device, rng_mode="philox", seed=0, use_graph=True,
                 world_size=1, rank=0, process_group=None, stream=None):
        """``world_size > 1``: synchronous data parallelism, one process per GPU over RCCL.  Every rank
        holds the same parameters, steps its own shard of the global batch (per-replica BatchNorm
        statistics, rank loss over the local pairs) and the five per-phase gradient arenas are averaged
        with one all-reduce each.  ``torch.distributed`` must already be initialised."""
        if not torch.cuda.is_available():
            raise RuntimeError("rankaae_amd.engine needs an MI355X (no CPU/PyTorch fallback for the training path)")
        _lib.load()
        self.cfg, self.device = dict(cfg), device
        # `tile_rows_mult`: the conv-network launches size their sample groups as for a batch this many times larger
        # (raae_tile_hint) -- for engines whose steps are launched T at a time by a TrialBatch (train_sc sets 4 there)
        self.tile_mult = int(self.cfg.get("tile_rows_mult", 1))
        ops.tile_hint(self.tile_mult)
        self.world_size, self.rank, self.pg = int(world_size), int(rank), process_group
        self.graph_ar = None
        # (`stream`: engines of a TrialBatch share one stream -- their batched step is one launch sequence on it)
        self._shared_stream = stream is not None
        self.stream = stream if stream is not None else torch.cuda.Stream(device=device)      # hipGraph capture is illegal on the null stream
        self.stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(self.stream):
            self._init(encoder, decoder, discriminator, cfg, device, rng_mode, seed, use_graph)

    def _init(self, encoder, decoder, discriminator, cfg, device, rng_mode, seed, use_graph):
        self.rng_mode, self.seed, self.use_graph = rng_mode, int(seed), use_graph
        self.enc_mod, self.dec_mod, self.dis_mod = encoder, decoder, discriminator
        self.arena = Arena([("disc", discriminator), ("enc", encoder), ("dec", decoder)], device)
        self.max_slab = 512
        self.G = torch.zeros(self.max_slab, self.arena.n, device=device)
        if self.world_size > 1:
            self.G_flat = torch.zeros(self.arena.n, device=device)          # all-reduce buffer
            self.seg_ones = torch.ones(self.arena.n // 64, dtype=torch.int16, device=device)
            import torch.distributed as dist
            self.comm_stream = torch.cuda.Stream(device=device)
            torch.cuda.current_stream().synchronize()
            with torch.cuda.stream(self.comm_stream):
                dist.broadcast(self.arena.P, src=0, group=self.pg)           # identical initial weights
            self.comm_stream.synchronize()
            # The per-phase all-reduce inside the step's hipGraph (own RCCL communicator), if it passes its
            # self-test on every rank; else the graph is cut at the collectives (_collective).  Measured on one
            # GPU: the cuts and event hops of the segmented path cost 45 us per phase.
            self.graph_ar = None
            if self.cfg.get("in_graph_allreduce", True) and dist.get_backend(self.pg) == "nccl":
                from .rccl import GraphAllReduce, agree
                ar = None
                try:
                    # (the constructor holds its own all-rank agreement before the collective communicator init:
                    # either every rank goes on or every rank raises)
                    ar = GraphAllReduce(dist.get_rank(self.pg), dist.get_world_size(self.pg), device, self.pg)
                except Exception:                                            # noqa: BLE001 -- fall back
                    ar = None
                # EVERY rank takes part in the agreement, also one whose communicator could not be created: a rank
                # that skipped it would leave the others waiting in the collective
                ok = ar.self_test() if ar is not None else agree(False, device, self.pg)
                self.graph_ar = ar if ok else None
        from .nets_conv import CompactNet   # local import: conv emitters live in their own module
        # build-only key `precision`: "fp32" (default, the reference's arithmetic and storage) | "bf16" (hidden
        # activations and dropout multipliers stored as bf16, everything else fp32: BASELINE configs[4])
        prec = str(cfg.get("precision", "fp32"))
        if prec not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', not {prec!r}")
        self.bf16 = prec == "bf16"
        if self.bf16 and cfg["ae_form"] != "FC":
            raise ValueError("precision: bf16 is implemented for ae_form: FC (BASELINE configs[4] is the dense network "
                             "on 512-point spectra; the conv networks are hard-wired to 256 points, SURVEY finding 5)")
        if cfg["ae_form"] == "FC":
            self.enc, self.dec = FCNet(encoder, "enc", self), FCNet(decoder, "dec", self)
        elif cfg["ae_form"] == "compact":
            self.enc, self.dec = CompactNet(encoder, "enc", self), CompactNet(decoder, "dec", self)
        else:
            raise ValueError(f"ae_form {cfg['ae_form']!r} is not reachable in the reference (SURVEY.md finding 4)")
        self.disc = DiscNet(discriminator, self)
        self.nstyle, self.n_aux = cfg["nstyle"], cfg["n_aux"]
        # rank loss under data parallelism: "local" = pairs inside each rank's shard (DDP semantics, no exchange);
        # "global" = all pairs of the global batch, i.e. the reference's loss on that batch (functions.py:63-77)
        self.rank_pairs_global = self.world_size > 1 and str(cfg.get("rank_loss_pairs", "local")) == "global"
        self.L = cfg["dim_in"]
        self._make_optimizers()
        self.steps_dev = torch.zeros(8, dtype=torch.int32, device=device)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=device)
        # [0]: step counter (advanced by the step's tick), [1]: seed, [2]: the step's two 32-bit dropout hash keys, stored
        # by the tick -- what the kernels that generate their own dropout multipliers read (raae_maskgen_t.keys);
        # rng_counter aliases [0] for raae_step_tick / raae_rng_fill
        self.rng_state = torch.tensor([0, self.seed, 0], dtype=torch.int64, device=device)    # counter, seed, hash keys
        self.rng_counter = self.rng_state[0:1]
        self.gen_state = self.rng_state       # where the dense kernels find the step's dropout hash keys (byte 16)
        # build-only key `inline_masks` (rng_mode "philox" only; default on): dropout multipliers are regenerated by
        # the kernels that apply them instead of written to and read from the random tape
        self.inline_masks = self.rng_mode == "philox" and bool(cfg.get("inline_masks", True))
        # experiment, build-only key `collapse_stats` (default off): the forwards whose backward
        # follows collapse their BatchNorm partial rows into one row for the three to five consumers of each statistic
        self.collapse_stats = bool(cfg.get("collapse_stats", False))
        self.collapse_min_rows = int(cfg.get("collapse_min_rows", 64))
        # build-only key `fused_step_begin` (default on): the head of a step -- counters, tape fill, batch gather and
        # spectral noise -- is one launch (raae_step_begin) instead of three
        self.fused_begin = bool(cfg.get("fused_step_begin", True))
        self.begin_ticket = torch.zeros(1, dtype=torch.int32, device=device)
        # build-only key `overlap_steps` (EXPERIMENT, default off; needs the device RNG, captured graphs, one GPU, a stream
        # of its own, a batch on the serial chain): the smoothness phase (trainer.py:189-200) updates only the decoder,
        # and once its encoder forward has run nothing in it reads encoder state; phase A of the NEXT step
        # (trainer.py:113-127) reads no decoder state.  The rest of the phase -- decoder forward, loss, decoder backward,
        # Adam -- is therefore deferred (`_tail`) and runs on a second stream beside the next step's phase A (three
        # single-stream graphs per step: `step`).  Same kernels, same operands, same order wherever there is a dependency:
        # bit for bit the plain step (tests/test_engine_gpu.py::test_overlapped_steps_are_bitwise_the_plain_steps).  What
        # the two would share is doubled: the random tape alternates between two buffers by step parity, the styles
        # and the dropout hash keys of the step are copied aside and the tail optimizer's step count advances at the
        # hand-over (raae_tail_prepare).  Every other public method first runs a pending tail (`finish`).
        # Off by default because it does not pay on this runtime: alone on two streams the two
        # graphs take max(177, 202) us, gated by the one event the tail needs they take 177 + 202.
        self.defer_tail = (bool(cfg.get("overlap_steps", False)) and self.rng_mode == "philox" and self.world_size == 1 and
                           bool(use_graph) and not self._shared_stream and self.fused_begin)
        self._tail = None                     # (plan, parity) of the step whose smoothness tail has not run yet
        self._step_no = 0
        self.tail_stream = torch.cuda.Stream(device=device) if self.defer_tail else None
        self._tail_ev = [torch.cuda.Event(), torch.cuda.Event()] if self.defer_tail else None
        self.tail_state = [torch.zeros(3, dtype=torch.int64, device=device) for _ in range(2)] if self.defer_tail else None
        self.alpha_dev = torch.zeros(1, device=device)
        self.loss_out = torch.zeros(8, device=device)
        self.taps = gaussian_taps(17, 3.0).tolist()
        self.bn_counts = {}
        self.plans = {}
        self.tape = None
        self._slab_notes = None
        self.train_spec = self.train_aux = self.perm = None
        self._perm_pin, self._perm_turn = None, 0
        self.phase_hook = None
        self.post_phase_hook = None
        self._capture = None
        n_side = int(self.cfg.get("side_streams", 3))
        self.side_streams = [torch.cuda.Stream(device=device) for _ in range(n_side)]
        self._side_i, self._side_used, self._events = 0, set(), []
        # Branches pay only when the kernels are long enough: every fork/join edge of a captured graph costs
        # about as much as a 10 us kernel (measured at B=256, conv networks: one serial chain 608 steps/s, three
        # side streams + the auxiliary stream 574; at B=4096 the branches win, 195 against 168; dense networks at
        # B=1024: 748 branched, 669 serial).
        # (With the weight gradients riding in the backward launches -- raae_block_bwd_b_wgrad -- the serial chain of
        # the conv networks also wins at 1024 rows: 428 against 417; branches from 2048: 305 against 288.  Round 3,
        # serial / branched: 512 rows 685 / 580, 1024 rows 553 / 512, 1536 rows 439 / 448: the threshold moved to 1536.)
        # (Dense networks, end of round 3: the serial chain with the discarded forwards paired into their neighbours' launches
        # wins at every size -- 919 against 873 steps/s at 1024 rows, 657 / 601 at 2048, 465 / 457 at 4096: no branches.)
        self.overlap_min_batch = int(self.cfg.get("overlap_min_batch", 1536 if self.cfg["ae_form"] == "compact" else 1 << 30))
        self._branch = True
        # one more stream for whole FORWARD chains whose result the step does not wait for (the two forwards
        # the reference runs only for their BatchNorm / RNG side effects): they run beside the critical chain
        # (pays off for the conv networks, +7 %; with the dense ones -- 5 us kernels -- the cross-stream
        # dependencies of the graph cost more than the overlap gains: 1018 -> 918 steps/s, so it is off there)
        overlap = self.cfg.get("overlap_unused_forwards", self.cfg["ae_form"] == "compact")
        self.aux_stream = torch.cuda.Stream(device=device) if overlap else None
        self.cursor_start, self.cursor_stride, self._cursor_primed = 0, None, False

    # -- optimizers: trainer.py:333-397 (only the five that ever step under gradient reversal)
    def _make_optimizers(self):
        c, r = self.cfg, self.arena.ranges
        lr = c["lr_base"]
        self.decoupled = {"AdamW": True, "Adam": False}[c["optimizer_name"]]
        default_wd = 0.01 if self.decoupled else 0.0
        betas_d = (c["dis_beta"] * 0.9, c["dis_beta"] * 0.009 + 0.99)
        spec = [("adversarial", r["disc"][0], r["enc"][1], c["lr_ratio_dis"] * lr, betas_d, default_wd),
                ("correlation", r["enc"][0], r["enc"][1], c["lr_ratio_Corr"] * lr, (0.9, 0.999), c["weight_decay"]),
                ("reconstruction", r["enc"][0], r["dec"][1], c["lr_ratio_Reconn"] * lr, (0.9, 0.999), c["weight_decay"]),
                ("mutual_info", r["enc"][0], r["dec"][1], c["lr_ratio_Mutual"] * lr, (0.9, 0.999), default_wd),
                ("smoothness", r["dec"][0], r["dec"][1], c["lr_ratio_Smooth"] * lr, (0.9, 0.999), c["weight_decay"])]
        self.opts = {n: OptState(i, n, lo, hi, l, b, 1e-8, wd, self.device) for i, (n, lo, hi, l, b, wd) in enumerate(spec)}

    # -- parameter-gradient kernels run on side streams (parallel branches of the captured graph):
    #    they are off the critical path of the data-gradient chain and only Adam needs their slabs.
    def side_stream(self):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if not self.side_streams or not self._branch:
                yield
                return
            s = self.side_streams[self._side_i % len(self.side_streams)]
            self._side_i += 1
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            s.wait_event(ev)
            self._events.append(ev)      # keep alive: destroying a recorded event mid-capture drops the edge on HIP
            self._side_used.add(s)
            with torch.cuda.stream(s):
                yield
        return ctx()

    def aux_branch(self):
        """Context: launches inside go to the auxiliary stream, ordered after everything submitted so far."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if self.aux_stream is None or not self._branch:
                yield
                return
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.aux_stream.wait_event(ev)
            self._events.append(ev)
            with torch.cuda.stream(self.aux_stream):
                yield
        return ctx()

    def join_aux(self):
        if self.aux_stream is None or not self._branch:
            return
        ev = torch.cuda.Event()
        ev.record(self.aux_stream)
        torch.cuda.current_stream().wait_event(ev)
        self._events.append(ev)

    def join_side_streams(self, keep=0):
        """The current stream waits for the side streams; ``keep`` > 0 leaves the ``keep`` most recently forked ones
        running (the side streams are used round-robin, so those are the last ``keep`` of the rotation)."""
        recent = set()
        if keep > 0 and self.side_streams:
            n = len(self.side_streams)
            recent = {self.side_streams[(self._side_i - 1 - j) % n] for j in range(min(keep, n - 1))}
        for s in list(self._side_used):
            if s in recent:
                continue
            ev = torch.cuda.Event()
            ev.record(s)
            torch.cuda.current_stream().wait_event(ev)
            self._events.append(ev)
            self._side_used.discard(s)
        if self._capture is None and len(self._events) > 4096:
            self._events.clear()

    # -- helpers used by the net emitters
    def gslab(self, p):
        return self.G[0, self.arena.off(p):]

    def note_slabs(self, params, ns):
        assert 1 <= ns <= self.max_slab
        if self._slab_notes is not None:
            for p in params:
                o = self.arena.off(p)
                self._slab_notes[o // 64:(o + p.numel() + 63) // 64] = ns

    def count_bn(self, bns):
        for bn in bns:
            self.bn_counts[id(bn)] = self.bn_counts.get(id(bn), 0) + 1

    @_on_stream
    def set_data(self, train_spec, train_aux):
        self.train_spec = torch.as_tensor(train_spec, dtype=torch.float32).contiguous().to(self.device)
        self.train_aux = torch.as_tensor(train_aux, dtype=torch.float32).contiguous().to(self.device)
        self.perm = torch.arange(len(self.train_spec), dtype=torch.int64, device=self.device)

    @_on_stream
    def set_epoch(self, perm, alpha, start=0, stride=None):
        """``start``/``stride``: this rank's rows of global batch i are perm[start + i*stride : +b]
        (single GPU: start 0, stride = b, i.e. consecutive batches)."""
        perm = torch.as_tensor(perm, dtype=torch.int64)
        if perm.is_cuda:
            self.perm.copy_(perm)
        else:
            # Through one of two pinned staging buffers, non-blocking: a copy from pageable memory makes the host wait
            # for every step already queued on the stream, i.e. drains the pipeline once per epoch (17 steps at batch
            # 4096).  A buffer is reused two epochs later; its event says the copy out of it has finished.
            if self._perm_pin is None or self._perm_pin[0][0].numel() != self.perm.numel():
                self._perm_pin = [(torch.empty(self.perm.numel(), dtype=torch.int64).pin_memory(), torch.cuda.Event())
                                  for _ in range(2)]
                self._perm_turn = 0
            buf, done = self._perm_pin[self._perm_turn]
            self._perm_turn ^= 1
            done.synchronize()
            buf.copy_(perm)
            self.perm.copy_(buf, non_blocking=True)
            done.record(torch.cuda.current_stream())
        self.cursor_start, self.cursor_stride = int(start), stride
        self._cursor_primed = False
        self._host_cursor = int(start)       # host mirror of the device row cursor (bounds check)
        self.alpha_dev.fill_(float(alpha))
        # (a fill kernel: `tensor[i] = 0.0` copies a host scalar from pageable memory, which makes the host wait for every
        # step queued on the stream -- it drained the pipeline once per epoch, and a host hiccup right behind it, in the
        # pinned-buffer copy above say, then cost the GPU its full length: the 10-25 % dips of 40-step windows at 4096 rows)
        k = LOSS_SLOTS["mi_accum"]
        self.loss_out[k:k + 1].zero_()

    @_on_stream
    def seek(self, start, stride=None):
        """Move the device row cursor inside the current epoch: the next step of ``b`` rows reads
        ``perm[start : start + b]`` and the following ones advance by ``stride`` (default: b)."""
        self.cursor_start, self.cursor_stride = int(start), stride
        self._cursor_primed = False
        self._host_cursor = int(start)

    @_on_stream
    def average_over_ranks(self, tensors):
        """In-place mean over the ranks of a list of small device tensors (BatchNorm running statistics before a
        validation pass).  The collective runs on the dedicated communication stream like every torch.distributed
        call of the engine: RCCL's work events must not be recorded on the engine's own stream, which captures
        hipGraphs (the process-group watchdog polls them with hipEventQuery)."""
        if self.world_size == 1 or not tensors:
            return
        flat = torch.cat([t.reshape(-1).float() for t in tensors])
        self._run_comm("mean", (flat,))
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()

    def close(self):
        """Release the private RCCL communicator (data-parallel runs).  ``Trainer.train`` calls it when the run ends, so
        that every rank destroys its communicator at the same point of the program.  Never while a hipGraph is being
        captured (``ncclCommDestroy`` synchronises and frees, which is illegal during capture): then it is a no-op and
        the communicator stays for a later call.  The captured steps hold nodes of that communicator: they are dropped,
        and a later ``step`` emits and captures again over ``torch.distributed`` between graph segments."""
        if self.graph_ar is None or self._capture is not None or ops.Graph.active > 0:
            return
        torch.cuda.synchronize(self.device)
        for P in self.plans.values():
            if hasattr(P, "graphs"):
                P.graphs = {}
            if getattr(P, "graph", None) is not None:
                P.graph, P.seen = None, 0
        self.graph_ar.close()
        self.graph_ar = None

    def release(self):
        """Give back what the engine holds on the device NOW -- captured graphs, workspaces, streams -- instead of when
        the cyclic collector finds it (the engine sits in reference cycles with its networks).  A process that runs many
        trials (``train_sc``) calls it after each; the engine cannot step afterwards.  The parameters (views of the
        arena) stay valid for the exported modules."""
        self.close()
        if self._capture is not None or ops.Graph.active > 0:
            return
        torch.cuda.synchronize(self.device)
        self.plans = {}
        self._events = []
        self.side_streams, self.aux_stream = [], None
        self.G = None
        self.tape = None

    def __del__(self):
        # the cyclic collector runs this at an arbitrary allocation point, possibly inside ANOTHER engine's capture
        # window in the same process (the next trial): close() refuses there, the communicator is then released with
        # the process
        try:
            self.close()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass

    # -- plan construction
    def plan(self, b):
        if b in self.plans:
            return self.plans[b]
        P = StepPlan()
        P.b = b
        P.stride = None
        c, dev, ns = self.cfg, self.device, self.nstyle
        bc = c["batch_size"]
        tape = Tape()
        P.tape = tape
        P.spec = torch.empty(b, self.L, device=dev)
        P.aux = torch.empty(b, self.n_aux, device=dev)
        P.enc, P.dec = self.enc.alloc(b), self.dec.alloc(b)
        P.disc = self.disc.alloc(bc, b)
        train = True
        # tape slots in the reference's consumption order
        # (device RNG with the fused step head: the spectral noise never touches the tape -- raae_step_begin generates it
        # where it adds it; the slot only takes its range of the Gaussian numbering)
        P.noise = tape.slot(b * self.L, 0, virtual=self.fused_begin and self.rng_mode == "philox")
        if P.noise >= 0:
            tape.draw("normal", P.noise, (b, self.L))
        P.m_enc, P.m_dec = [], []
        P.m_enc.append(self.enc.mask_slots(tape, b))            # trainer.py:113
        P.m_dec.append(self.dec.mask_slots(tape, b))            # :114
        P.sl_disc = self.disc.tape_slots(tape, bc, b, train)    # phase A
        P.m_enc.append(self.enc.mask_slots(tape, b))            # phase B :154
        P.m_enc.append(self.enc.mask_slots(tape, b))            # phase C :165
        P.m_dec.append(self.dec.mask_slots(tape, b))
        P.m_enc.append(self.enc.mask_slots(tape, b))            # phase D :176
        P.z_sample = tape.slot(b * ns, 0)
        tape.draw("normal", P.z_sample, (b, ns))
        P.m_dec.append(self.dec.mask_slots(tape, b))
        P.m_enc.append(self.enc.mask_slots(tape, b))
        P.n_draws_no_smooth = len(tape.draws)
        P.m_enc.append(self.enc.mask_slots(tape, b))            # phase E :191
        P.m_dec.append(self.dec.mask_slots(tape, b))
        tape.finalize(dev)
        tape.bufs = [tape.buf, torch.zeros_like(tape.buf)] if self.defer_tail else [tape.buf]
        P.styles_tail = torch.empty(b, ns, device=dev)
        P.rank_work = torch.empty(ops.rank_loss_work_bytes(b, self.n_aux), dtype=torch.uint8, device=dev)
        if self.rank_pairs_global:
            P.aux_all = torch.empty(self.world_size * b, self.n_aux, device=dev)
            P.z_all = torch.empty(self.world_size * b, ns, device=dev)
            P.rank_totals = torch.zeros(64, dtype=torch.float64, device=dev)
        P.dstyles = torch.empty(b, ns, device=dev)
        P.dspec = torch.empty(b, self.L, device=dev)
        P.dout = torch.empty(b, self.L, device=dev)
        P.lpart = torch.zeros(RAAE_MAX_PARTS, dtype=torch.float64, device=dev)
        P.ticket = torch.zeros(1, dtype=torch.int32, device=dev)      # arrival counter of the in-kernel loss sums
        P.seg = {n: torch.zeros(self.arena.n // 64, dtype=torch.int16, device=dev) for n in OPT_NAMES}
        P.max_slab = {n: 0 for n in OPT_NAMES}
        P.graphs = {}
        self.plans[b] = P
        return P

    # -- emit the step program (eagerly or under capture)
    def _adam(self, P, name, notes_host):
        o = self.opts[name]
        if notes_host is not None:
            P.seg[name].copy_(torch.from_numpy(notes_host))
            P.max_slab[name] = int(notes_host.max())      # host-side hint for the Adam kernel's lane split
        self.join_side_streams()
        if self.phase_hook is not None:      # debugging / parity tests: gradients before the update
            self._host_hook(self.phase_hook, name, P)
        lo, n = o.lo, o.hi - o.lo
        if self.world_size > 1:
            # flat gradient -> RCCL mean over ranks -> Adam on the averaged single slab
            ops.slab_reduce(self.G[0, lo:], self.arena.n, P.seg[name][lo // 64:], n, self.G_flat[lo:], P.max_slab[name])
            self._collective(self.G_flat[lo:o.hi])
            ops.adam_step(self.arena.P[lo:], o.m, o.v, self.G_flat[lo:], self.arena.n, self.seg_ones[lo // 64:], n,
                          o.hyper, self.steps_dev[o.index:], self.decoupled, 1)
        else:
            ops.adam_step(self.arena.P[lo:], o.m, o.v, self.G[0, lo:], self.arena.n, P.seg[name][lo // 64:], n,
                          o.hyper, self.steps_dev[o.index:], self.decoupled, P.max_slab[name])
        if self.post_phase_hook is not None:  # parity tests: teacher forcing at phase granularity
            self._host_hook(self.post_phase_hook, name, P)

    def _host_hook(self, fn, name, P):
        """A host callback between two launches (parity tests).  Eager emission: call it now.  Under capture: the
        graph is cut here -- the segment so far ends, the callback becomes an item of the replay list, a new
        segment begins (all branches were joined just before, so the cut is legal)."""
        if self._capture is None:
            fn(name, P)
            return
        g = self._capture["cur"]
        g.end()
        self._capture["items"] += [g, lambda: fn(name, P)]
        g = ops.Graph()
        g.begin()
        self._capture["cur"] = g

    def _run_comm(self, kind, bufs):
        """One collective through ``torch.distributed`` on the dedicated communication stream.  RCCL's work events
        must never be recorded on a stream that later captures a hipGraph: the process-group watchdog polls them
        with hipEventQuery, which HIP rejects for an event last recorded on a capturing stream."""
        import torch.distributed as dist
        from .parallel import allreduce_mean_
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            if kind == "mean":
                allreduce_mean_(bufs[0], self.pg)
            elif kind == "sum":
                dist.all_reduce(bufs[0], op=dist.ReduceOp.SUM, group=self.pg)
            else:                                   # gather: dst[r*n:(r+1)*n] = src of rank r
                src, dst = bufs
                dist.all_gather(list(dst.view(self.world_size, -1).unbind(0)), src.reshape(-1), group=self.pg)
            ev2 = torch.cuda.Event()
            ev2.record(self.comm_stream)
        cur.wait_event(ev2)

    def _all_reduce(self, buf):
        self._run_comm("mean", (buf,))

    def _collective(self, buf, kind="mean", dst=None):
        """``kind``: "mean" (gradient arenas), "sum" (float64 pair totals), "gather" (buf -> dst).  With the private
        RCCL communicator the call is a node of the graph being captured (or an eager launch on this stream).
        Otherwise -- eager emission: run it now through torch.distributed; under capture: close the current graph
        segment, remember the collective, open the next segment (RCCL calls stay outside the hipGraphs)."""
        bufs = (buf,) if dst is None else (buf, dst)
        if self.graph_ar is not None:
            {"mean": self.graph_ar.mean_, "sum": self.graph_ar.sum_, "gather": self.graph_ar.gather_}[kind](*bufs)
        elif self._capture is None:
            self._run_comm(kind, bufs)
        else:
            g = self._capture["cur"]
            g.end()
            self._capture["items"] += [g, lambda: self._run_comm(kind, bufs)]
            g = ops.Graph()
            g.begin()
            self._capture["cur"] = g

    def _rank_loss(self, P, styles):
        """Phase B's loss and d(loss)/d(styles).  Data parallel with ``rank_loss_pairs: global``: the ranks exchange
        their [b, n_aux] descriptors and styles (all-gather), every rank pairs ITS rows with all W*b rows, the
        per-descriptor pair counts and sums meet in one 512-byte all-reduce, and the row-local gradient is exact;
        it is scaled by W because the parameter gradients are averaged over the ranks afterwards."""
        c, b, ns, lo = self.cfg, P.b, self.nstyle, self.loss_out
        if not self.rank_pairs_global:
            ops.rank_loss_fwd_bwd(P.aux, self.n_aux, styles, ns, b, self.n_aux, c["kendall_activation"], P.rank_work,
                                  lo[1:2], P.dstyles)
            return
        W = self.world_size
        self._collective(P.aux, "gather", P.aux_all)
        self._collective(styles, "gather", P.z_all)
        ops.rank_rows_pairs(P.aux_all, self.n_aux, P.z_all, ns, W * b, self.rank * b, b, self.n_aux, P.rank_work,
                            P.rank_totals)
        self._collective(P.rank_totals, "sum")
        ops.rank_rows_finish(P.rank_totals, W * b, b, self.n_aux, c["kendall_activation"], float(W), P.rank_work,
                             lo[1:2], P.dstyles, ns)

    def _begin_phase(self, record):
        self._slab_notes = np.zeros(self.arena.n // 64, dtype=np.int16) if record else None

    def emit_step(self, P, smooth, record, parity=0, defer=False, tail_join=None):
        """``record``: first (eager) emission -- slab counts are recorded into the Adam segment
        tables.  Under graph capture ``record`` is False (no host->device copies).  ``parity``: which of the plan's
        tape buffers this step fills and reads.  ``defer`` (`overlap_steps`): the smoothness phase stops behind its
        encoder forward and hands the rest over (``_emit_tail`` runs it later); ``tail_join`` (under capture): cut the
        graph behind phase A -- the PREVIOUS step's tail runs beside the first part and is waited for before the
        second."""
        c, b, ns = self.cfg, P.b, self.nstyle
        self.tape = P.tape
        tape = P.tape
        tape.buf = tape.bufs[parity % len(tape.bufs)]
        self.gen_state = self.rng_state
        self._branch = b >= self.overlap_min_batch
        # (deferred tail: its optimizer's step count advances at the hand-over, not here -- the previous step's tail may
        # still be reading it)
        mask_bits = 0b01111 | (0b10000 if (smooth and not defer) else 0)
        stride = self.cursor_stride if self.cursor_stride is not None else b
        if P.stride is None:
            P.stride = stride
        assert P.stride == stride, "cursor stride is baked into the captured graph of this batch size"
        if self.fused_begin:
            # tick + tape fill + gather in ONE launch (was three: 19 us of the 256-row step)
            philox = self.rng_mode == "philox"
            ops.step_begin(self.steps_dev, 5, mask_bits, self.rng_state, self.cursor, stride, self.begin_ticket,
                           self.train_spec, self.train_aux, self.perm, b, self.L, self.n_aux, float(c["spec_noise"]),
                           None if philox else tape.view(P.noise, b, self.L), (-P.noise - 1) if philox else 0,
                           P.spec, P.aux, tape if philox else None)
        else:
            ops.step_tick(self.steps_dev, 5, mask_bits, self.rng_counter, self.cursor, stride)
            if self.rng_mode == "philox":
                ops.rng_fill(tape.buf, tape.seg_desc, tape.seg_scale, len(tape.segs), tape.total, self.seed,
                             self.rng_counter)
            ops.gather_batch(self.train_spec, self.train_aux, self.perm, self.cursor, tape.view(P.noise, b, self.L),
                             float(c["spec_noise"]), b, self.L, self.n_aux, P.spec, P.aux)
        enc, dec, E, D = self.enc, self.dec, P.enc, P.dec
        lo = self.loss_out

        def will_backprop(*nets):       # `collapse_stats`: the next forward of these networks is followed by its backward
            for net in (enc, dec):
                net.collapse = self.collapse_stats and net in nets
        # trainer.py:113-114
        will_backprop(enc)
        styles = enc.forward(E, P.spec, P.m_enc[0])
        will_backprop()
        # The reference discards this decoder output (BatchNorm statistics and RNG draws are its only effects).
        # Serial chain: it is deferred to phase B, where it runs in lockstep with the encoder forward (one launch
        # per pair of block kernels) -- phase A updates neither the decoder nor, before that forward's last
        # kernel, `styles`, so the results are bit for bit those of running it here
        # (tests/test_engine_gpu.py::test_paired_forwards_change_nothing).  Branched graph: beside phase A.
        # Only legal when the encoder's forward hands control back (one yield per launch) before it overwrites
        # `styles`, so that the deferred decoder forward's first launch -- the only one that reads `styles` -- is
        # issued first: true for the dense networks and for fused residual blocks, not for the per-layer conv path.
        pair = ((not self._branch) and bool(c.get("pair_unused_forwards", True)) and
                getattr(enc, "pairable", False) and getattr(dec, "pairable", False))
        if not pair:
            with self.aux_branch():
                dec.forward(D, styles, P.m_dec[0])
        # ---- phase A: adversarial (trainer.py:117-127)
        self._begin_phase(record)
        dst = self.disc.forward_backward(P.disc, P.sl_disc, styles, lo[0:1])
        enc.backward(E, P.spec, P.m_enc[0], dst)
        self.join_aux()
        self._adam(P, "adversarial", self._slab_notes)
        if tail_join:                    # the previous step's tail (decoder only) runs beside everything above: the
            g = self._capture["cur"]     # step is TWO graphs, and the main stream waits for the tail between them
            g.end()
            self._capture["items"].append(g)
            g = ops.Graph()
            g.begin()
            self._capture["cur"] = g
        # ---- phase B: rank correlation (:153-161)
        self._begin_phase(record)
        will_backprop(enc)
        if pair:
            styles, _ = enc.forward_pair(enc.forward_steps(E, P.spec, P.m_enc[1]), dec.forward_steps(D, styles, P.m_dec[0]))
        else:
            styles = enc.forward(E, P.spec, P.m_enc[1])
        will_backprop()
        self._rank_loss(P, styles)
        enc.backward(E, P.spec, P.m_enc[1], P.dstyles)
        self._adam(P, "correlation", self._slab_notes)
        # ---- phase C: reconstruction (:164-172)
        self._begin_phase(record)
        will_backprop(enc, dec)
        styles = enc.forward(E, P.spec, P.m_enc[2])
        out = dec.forward(D, styles, P.m_dec[1])
        will_backprop()
        ops.recon_loss_fwd_bwd(P.spec, out, b, self.L, c["use_flex_spec_target"], P.lpart, P.dout,
                               fin=(1.0, lo, 2, -1, P.ticket))
        left = dec.backward(D, styles, P.m_dec[1], P.dout, P.dstyles, keep_pending=True)
        enc.backward(E, P.spec, P.m_enc[2], P.dstyles, pending=left)
        self._adam(P, "reconstruction", self._slab_notes)
        # ---- phase D: mutual information (:175-186)
        self._begin_phase(record)
        z_s = tape.view(P.z_sample, b, ns)
        will_backprop(dec)               # (the encoder forward of this pair is the one the reference discards)
        if pair:
            # the encoder forward whose result the reference does not use (BN stats + RNG only) and the decoder
            # forward, which only needs z_sample, in lockstep: one launch per pair of block kernels
            _, out = enc.forward_pair(enc.forward_steps(E, P.spec, P.m_enc[3]), dec.forward_steps(D, z_s, P.m_dec[2]))
        else:
            with self.aux_branch():                 # ... or on the auxiliary stream, beside the decoder forward
                enc.forward(E, P.spec, P.m_enc[3])
            out = dec.forward(D, z_s, P.m_dec[2])
            self.join_aux()
        will_backprop(enc)
        z_rec = enc.forward(E, out, P.m_enc[4])
        will_backprop()
        ops.mse_fwd_bwd(z_rec, z_s, b * ns, P.lpart, P.dstyles, fin=(1.0, lo, 3, 5, P.ticket))
        left = enc.backward(E, out, P.m_enc[4], P.dstyles, P.dspec, keep_pending=True)
        dec.backward(D, z_s, P.m_dec[2], P.dspec, None, pending=left)
        self._adam(P, "mutual_info", self._slab_notes)
        # ---- phase E: smoothness (:189-200); encoder gradients are discarded by the reference
        if smooth and defer:
            styles = enc.forward(E, P.spec, P.m_enc[5])        # (its gradients are discarded: no backward)
            ops.tail_prepare(styles, P.styles_tail, b * ns, self.steps_dev[4:], self.rng_state, self.tail_state[parity & 1])
        elif smooth:
            self._begin_phase(record)
            styles = enc.forward(E, P.spec, P.m_enc[5])        # (its gradients are discarded: no backward)
            will_backprop(dec)
            out = dec.forward(D, styles, P.m_dec[3])
            will_backprop()
            ops.smooth_loss_fwd_bwd(out, b, self.L, self.taps, P.lpart, P.dout, fin=(1.0, lo, 4, -1, P.ticket))
            dec.backward(D, styles, P.m_dec[3], P.dout, None)
            self._adam(P, "smoothness", self._slab_notes)
        self._slab_notes = None

    def _emit_tail(self, P, parity):
        """The deferred rest of a step's smoothness phase (``emit_step(defer=True)``): decoder forward on the styles set
        aside, loss, decoder backward, Adam -- with the tape buffer and the dropout hash keys of ITS step."""
        b = P.b
        tape = P.tape
        saved = (self.tape, tape.buf, self.gen_state)
        self.tape = tape
        tape.buf = tape.bufs[parity % len(tape.bufs)]
        self.gen_state = self.tail_state[parity & 1]
        enc, dec, D = self.enc, self.dec, P.dec
        enc.collapse, dec.collapse = False, self.collapse_stats
        out = dec.forward(D, P.styles_tail, P.m_dec[3])
        dec.collapse = False
        ops.smooth_loss_fwd_bwd(out, b, self.L, self.taps, P.lpart, P.dout, fin=(1.0, self.loss_out, 4, -1, P.ticket))
        dec.backward(D, P.styles_tail, P.m_dec[3], P.dout, None)
        self._adam(P, "smoothness", None)
        self.tape, tape.buf, self.gen_state = saved

    def _run_tail(self):
        """Run a pending tail now, eagerly, on the current stream (before anything that reads or replaces what it
        touches: losses, validation, exports, another batch size, the end of a run)."""
        t, self._tail = self._tail, None
        if t is not None:
            self._emit_tail(*t)

    @_on_stream
    def finish(self):
        """Complete the last step (its deferred smoothness tail, if any); every public method but ``step`` does so
        itself -- call it before timing the end of a run of steps."""

    def _pre_step(self, b, smooth):
        """Host side of a step before anything is launched: plan, bounds check, device cursor priming, host tape."""
        if b < 2:
            # a one-row last batch: the reference's training-mode BatchNorm1d(nstyle) raises exactly this
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             f"torch.Size([{b}, {self.nstyle}])")
        P = self.plan(b)
        stride = self.cursor_stride if self.cursor_stride is not None else b
        if self._host_cursor + b > len(self.train_spec):
            raise RuntimeError(f"epoch exhausted: rows [{self._host_cursor}, {self._host_cursor + b}) exceed the "
                               f"{len(self.train_spec)}-row training split; call set_epoch() first")
        self._host_cursor += stride
        if not self._cursor_primed:
            # the tick adds `stride` BEFORE the gather, which reads rows [cursor - b, cursor)
            self.cursor.fill_(self.cursor_start + b - stride)
            self._cursor_primed = True
        if self.rng_mode == "host":
            P.tape.draws, saved = (P.tape.draws if smooth else P.tape.draws[:P.n_draws_no_smooth]), P.tape.draws
            P.tape.fill_host()
            P.tape.draws = saved
        return P

    @_on_stream
    def step(self, b, smooth=True):
        """Run one training step on the next ``b`` rows of the epoch permutation."""
        P = self._pre_step(b, smooth)
        key = bool(smooth)
        parity = self._step_no & 1
        self._step_no += 1
        # (only where the decoder forward the reference discards rides with phase B's encoder forward: run inline in
        # phase A it would share the decoder's workspace and running statistics with the tail beside it)
        defer = (self.defer_tail and b < self.overlap_min_batch and self.phase_hook is None and
                 self.post_phase_hook is None and bool(self.cfg.get("pair_unused_forwards", True)) and
                 getattr(self.enc, "pairable", False) and getattr(self.dec, "pairable", False))
        if self._tail is not None and (not defer or self._tail[0] is not P or key not in P.graphs):
            self._run_tail()               # another batch size / an eager step follows: nothing to run beside
        if key not in P.graphs:
            # first call: eager emission (records slab counts, sets kernel attributes) ...
            self.emit_step(P, smooth, record=True, parity=parity)
            P.graphs[key] = None
            return
        if self.use_graph and defer:
            # `overlap_steps`: one graph per (smoothness phase or not, tape parity, a tail of the previous step to run
            # beside this step's phase A or not); this step's own tail stays pending
            # Three single-stream graphs (a captured graph with branches is re-submitted node by node on the host at every
            # launch -- ~0.8 ms for these 150 nodes, the step itself; single-stream graphs replay from prepared packets):
            # H1 = head up to the end of phase A, H2 = the rest of the head, T = a tail.  T of the previous step goes to
            # the tail stream behind an event of this stream, H2 waits for its end.
            tail = self._tail
            if (key, parity) not in P.graphs:
                # both parities at once (a capture takes tens of milliseconds: better now than an epoch into a run)
                for p_ in (0, 1):
                    if (key, p_) not in P.graphs:
                        P.graphs[(key, p_)] = self._capture_overlapped(P, smooth, p_)
                    if smooth and ("tail", p_) not in P.graphs:
                        P.graphs[("tail", p_)] = self._capture_tail(P, p_)
            h1, h2 = P.graphs[(key, parity)]
            main = torch.cuda.current_stream()
            if tail is not None:
                self._tail_ev[0].record(main)
                self.tail_stream.wait_event(self._tail_ev[0])
                with torch.cuda.stream(self.tail_stream):
                    P.graphs[("tail", tail[1])].launch()
                    self._tail_ev[1].record(self.tail_stream)
            h1.launch()
            if tail is not None:
                main.wait_event(self._tail_ev[1])
            h2.launch()
            self._tail = (P, parity) if smooth else None
            self._count_bn_step(smooth)
            return
        if self.use_graph and P.graphs[key] is None:
            # ... second call: capture; the capture itself does not execute, so launch it right away
            bn_saved = dict(self.bn_counts)
            g = ops.Graph()
            g.begin()
            self._capture = {"cur": g, "items": []}
            self.emit_step(P, smooth, record=False, parity=parity)
            self._capture["cur"].end()
            P.graphs[key] = self._capture["items"] + [self._capture["cur"]]
            self._capture = None
            self.bn_counts = bn_saved
        if self.use_graph:
            for item in P.graphs[key]:
                if isinstance(item, ops.Graph):
                    item.launch()
                else:
                    item()
            self._count_bn_step(smooth)
        else:
            self.emit_step(P, smooth, record=False, parity=parity)

    def _capture_overlapped(self, P, smooth, parity):
        """The two graphs of one `overlap_steps` step: the head up to the end of phase A, and the rest of the head; the
        step's own tail is left out."""
        bn_saved = dict(self.bn_counts)
        g = ops.Graph()
        g.begin()
        self._capture = {"cur": g, "items": []}
        self.emit_step(P, smooth, record=False, parity=parity, defer=True, tail_join=True)
        self._capture["cur"].end()
        items = self._capture["items"] + [self._capture["cur"]]
        self._capture = None
        self.bn_counts = bn_saved
        assert len(items) == 2 and all(isinstance(i, ops.Graph) for i in items), "a host callback cut the graph of an overlapped step"
        return tuple(items)

    def _capture_tail(self, P, parity):
        bn_saved = dict(self.bn_counts)
        g = ops.Graph()
        g.begin()
        self._capture = {"cur": g, "items": []}
        self._emit_tail(P, parity)
        g.end()
        assert not self._capture["items"]
        self._capture = None
        self.bn_counts = bn_saved
        return g

    # -- roofline probe (bench.py): HIP-event timing of every kernel family the step launches
    @_on_stream
    def roofline_probe(self, b, peak_gbs, reps=3, top=3, detail=False):
        """``reps`` eager steps with ``ops.PROBE`` armed: every launch of the step is followed by ten identical
        launches replayed from a small hipGraph between two HIP events on the launching stream.  Returns the ``top``
        kernel families by share of the summed kernel time, each with its algorithmic bytes per launch (ops.block_bytes
        and friends: SURVEY 8d's per-sample figure x the batch), average launch duration and fraction of the HBM peak;
        the first one is the step's dominant kernel.

        The replays are NOT idempotent (accumulating data-gradient kernels, ``out[acc_slot] += v`` loss finishers, Adam
        steps on the results): the engine's trainable state -- parameters, Adam moments and step counts, BatchNorm
        running statistics, loss slots, RNG counter -- is snapshotted before the probe and restored after it, so the
        engine can go on training from where it was."""
        saved_graph = self.use_graph
        self.use_graph = False
        torch.cuda.synchronize()
        bufs = [self.arena.P, self.steps_dev, self.loss_out, self.rng_counter, self.cursor] + \
               [t for o in self.opts.values() for t in (o.m, o.v)] + \
               [b_ for mod in (self.enc_mod, self.dec_mod, self.dis_mod) for b_ in mod.buffers()]
        snap = [t.clone() for t in bufs]
        host = (dict(self.bn_counts), self._host_cursor, self._cursor_primed, self.cursor_start, self.cursor_stride)
        self.set_epoch(self.perm.clone(), float(self.alpha_dev))
        self.step(b, smooth=True)                  # plan + slab tables exist before the probe arms
        ops.PROBE = ops.Probe(PROBE_REPS, detail)
        try:
            for _ in range(reps):
                self.step(b, smooth=True)
            torch.cuda.synchronize()
            fams = ops.PROBE.summary()
        finally:
            ops.PROBE = None
            self.use_graph = saved_graph
            for t, v in zip(bufs, snap):
                t.copy_(v)
            self.bn_counts, self._host_cursor, self._cursor_primed, self.cursor_start, self.cursor_stride = host
            torch.cuda.synchronize()
        total = sum(f["total_us"] for f in fams.values())
        rows = []
        for name, f in sorted(fams.items(), key=lambda kv: -kv[1]["total_us"]):
            ach = f["bytes"] / (f["avg_us"] * 1e-6) / 1e9
            rows.append({"kernel": name, "share_of_kernel_time": round(f["total_us"] / total, 4),
                         "launches_per_step": round(f["launches"] / reps, 2), "avg_launch_us": round(f["avg_us"], 2),
                         "algorithmic_bytes_per_launch": int(f["bytes"]), "achieved": round(ach, 1), "unit": "GB/s",
                         "frac": round(ach / peak_gbs, 5)})
        lead = rows[0]
        return {"bound": "hbm", "kernel": lead["kernel"], "achieved": lead["achieved"], "peak": peak_gbs, "unit": "GB/s",
                "frac": lead["frac"], "traffic": None, "launches_per_step": lead["launches_per_step"],
                "avg_launch_us": lead["avg_launch_us"], "algorithmic_bytes_per_launch": lead["algorithmic_bytes_per_launch"],
                "batch": b, "kernel_time_us_per_step": round(total / reps, 1), "top_kernels": rows[:top],
                "all_kernels": {r["kernel"]: [r["share_of_kernel_time"], r["avg_launch_us"], r["frac"]] for r in rows}}

    @_on_stream_io
    def reconstruct(self, spec):
        """Eval-mode ``styles = Encoder(spec)``, ``spec_out = Decoder(styles)`` for ``[n, L]`` device rows --
        the latent-space export of the reference's report tool (sc/report/analysis_new.py:94-129)."""
        n = spec.shape[0]
        key = ("recon", n)
        if key not in self.plans:
            R = StepPlan()
            R.enc, R.dec = self.enc.alloc(n), self.dec.alloc(n)
            self.plans[key] = R
        R = self.plans[key]
        z = self.enc.forward(R.enc, spec, None, train=False)
        out = self.dec.forward(R.dec, z, None, train=False)
        return z.clone(), out.clone()

    @_on_stream_io
    def decode(self, styles, n_sampling=1):
        """Eval-mode ``Decoder(styles)`` for ``[n, nstyle]`` device rows; with ``n_sampling > 1`` the rows are
        ``[n / n_sampling][n_sampling]`` and the mean spectrum of each group is returned -- the decoder sweeps of
        the reference's report (sc/report/analysis.py:68-86)."""
        n = styles.shape[0]
        assert styles.shape[1] == self.nstyle and n % n_sampling == 0
        key = ("decode", n)
        if key not in self.plans:
            R = StepPlan()
            R.dec = self.dec.alloc(n)
            self.plans[key] = R
        out = self.dec.forward(self.plans[key].dec, styles.contiguous(), None, train=False)
        if n_sampling == 1:
            return out.clone()
        mean = torch.empty(n // n_sampling, self.L, device=self.device)
        ops.group_mean(out, n // n_sampling, n_sampling, self.L, mean)
        return mean

    def phase_gradient(self, P, name):
        """Flat gradient (fixed-order slab sum) of optimizer ``name``'s arena range -- what the
        fused Adam kernel consumes.  For tests and debugging."""
        o = self.opts[name]
        seg = P.seg[name][o.lo // 64:o.hi // 64].long().repeat_interleave(64)
        g = torch.zeros(o.hi - o.lo, device=self.device)
        for s in range(int(seg.max())):
            g += torch.where(seg > s, self.G[s, o.lo:o.hi], torch.zeros_like(g))
        return g

    @_on_stream
    def load_optimizer_state(self, name, params, torch_optimizer):
        """Copy ``exp_avg`` / ``exp_avg_sq`` / ``step`` of a ``torch.optim.Adam(W)`` whose
        parameters correspond, in order, to ``params`` (this engine's parameters)."""
        o = self.opts[name]
        theirs = [p for grp in torch_optimizer.param_groups for p in grp["params"]]
        assert len(theirs) == len(params)
        step = 0
        for mine, other in zip(params, theirs):
            st = torch_optimizer.state.get(other, {})
            off = self.arena.off(mine) - o.lo
            n = mine.numel()
            if st:
                o.m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                o.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = int(st["step"])
            else:
                o.m[off:off + n].zero_()
                o.v[off:off + n].zero_()
        self.steps_dev[o.index] = step
        for grp in torch_optimizer.param_groups:
            o.lr = float(grp["lr"])
        o.push()

    def _count_bn_step(self, smooth):
        n_enc, n_dec = (6, 4) if smooth else (5, 3)
        for bn in self.enc.bn_modules:
            self.bn_counts[id(bn)] = self.bn_counts.get(id(bn), 0) + n_enc
        for bn in self.dec.bn_modules:
            self.bn_counts[id(bn)] = self.bn_counts.get(id(bn), 0) + n_dec

    @_on_stream
    def sync_bn_counters(self):
        """Write ``num_batches_tracked`` (host-side count of train-mode forwards) into the modules."""
        for net in (self.enc, self.dec):
            for bn in net.bn_modules:
                bn.num_batches_tracked.fill_(self.bn_counts.get(id(bn), 0))

    @_on_stream
    def losses(self):
        v = self.loss_out.cpu().tolist()
        return {k: v[i] for k, i in LOSS_SLOTS.items()}

    # -- validation (trainer.py:206-268): eval-mode forward of the whole validation set
    @_on_stream
    def validate(self, val_spec, val_aux, rng_host=True, _phase=None):
        """``_phase`` (TrialBatch): "emit" -- launch the validation program eagerly and return ``(z, None)`` without
        reading anything back; "read" -- only read the results of a program that has run."""
        c, dev, ns = self.cfg, self.device, self.nstyle
        nv, bc = val_spec.shape[0], c["batch_size"]
        key = ("val", nv)
        # Data parallel: the O(n_val^2) rank loss is SHARDED -- every rank pairs its rows [row0, row0 + nrows) with all
        # n_val rows (raae_rank_rows_pairs), the per-descriptor counts and sums meet in one 512-byte float64 all-reduce,
        # and raae_rank_rows_finish forms the loss, identical on every rank and equal to the replicated computation to
        # the order of the float64 sums (47 ms -> 47 / W ms at n_val = 150 k).  The O(n_val) parts (forwards, the four
        # other losses, the style metrics, which need every row's styles) stay replicated.
        shard = self.world_size > 1 and bool(c.get("shard_validation", True))
        if key not in self.plans:
            V = StepPlan()
            V.enc, V.dec = self.enc.alloc(nv), self.dec.alloc(nv)
            V.enc2 = self.enc.alloc(nv)
            V.disc = self.disc.alloc(bc, nv)
            V.tape = Tape()
            V.z_sample = V.tape.slot(nv * ns, 0)
            V.tape.draw("normal", V.z_sample, (nv, ns))
            V.sl_disc = self.disc.tape_slots(V.tape, bc, nv, train=False)
            V.tape.finalize(dev)
            V.rank_work = torch.empty(ops.rank_loss_work_bytes(nv, self.n_aux), dtype=torch.uint8, device=dev)
            V.rank_totals = torch.zeros(64, dtype=torch.float64, device=dev)
            V.lpart = torch.zeros(RAAE_MAX_PARTS, dtype=torch.float64, device=dev)
            V.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
            V.out = torch.zeros(8, device=dev)
            V.metrics = StyleMetrics(nv, ns, dev)
            self.plans[key] = V
        V = self.plans[key]
        self._val_plan = V
        if _phase == "read":
            v = V.out.cpu().tolist()
            return V.z, {k: v[i] for k, i in LOSS_SLOTS.items() if k != "mi_accum"}
        self.tape = V.tape
        if self.rng_mode == "host":
            torch.empty((), dtype=torch.int64).random_()     # the val DataLoader iterator's _base_seed draw
            V.tape.fill_host()

        def emit():
            if self.rng_mode != "host":
                ops.rng_fill(V.tape.buf, V.tape.seg_desc, V.tape.seg_scale, len(V.tape.segs), V.tape.total,
                             self.seed ^ 0x5EED, self.rng_counter)
            z = self.enc.forward(V.enc, val_spec, None, train=False)
            out = self.dec.forward(V.dec, z, None, train=False)
            ops.recon_loss_fwd_bwd(val_spec, out, nv, self.L, False, V.lpart, None, fin=(1.0, V.out, 2, -1, V.ticket))
            if shard:
                per = (nv + self.world_size - 1) // self.world_size
                row0 = min(self.rank * per, nv)
                nrows = min(per, nv - row0)
                if nrows > 0:
                    ops.rank_rows_pairs(val_aux, self.n_aux, z, ns, nv, row0, nrows, self.n_aux, V.rank_work, V.rank_totals)
                else:                       # more ranks than validation rows: this rank adds nothing
                    V.rank_totals.zero_()
                self._collective(V.rank_totals, "sum")
                ops.rank_rows_finish(V.rank_totals, nv, max(nrows, 1), self.n_aux, c["kendall_activation"], 1.0,
                                     V.rank_work, V.out[1:2], None, ns)
            else:
                ops.rank_loss_fwd_bwd(val_aux, self.n_aux, z, ns, nv, self.n_aux, c["kendall_activation"], V.rank_work,
                                      V.out[1:2], None)
            ops.smooth_loss_fwd_bwd(out, nv, self.L, self.taps, V.lpart, None, fin=(1.0, V.out, 4, -1, V.ticket))
            z_s = V.tape.view(V.z_sample, nv, ns)
            out2 = self.dec.forward(V.dec, z_s, None, train=False)
            z_rec = self.enc.forward(V.enc2, out2, None, train=False)
            ops.mse_fwd_bwd(z_rec, z_s, nv * ns, V.lpart, None, fin=(1.0, V.out, 3, -1, V.ticket))
            self.disc.forward_backward(V.disc, V.sl_disc, z, V.out[0:1], train=False)
            V.metrics.launch(z)      # Shapiro-Wilk W per style, Spearman rho per pair (trainer.py:286-292)
            return z

        if _phase == "emit":
            V.z = emit()
            return V.z, None
        # like the training step: eager once, captured on the second call, replayed afterwards (the inputs must
        # then be the same device tensors: the trainer validates on one resident split)
        same = getattr(V, "inputs", None) == (val_spec.data_ptr(), val_aux.data_ptr())
        if shard and self.graph_ar is None:
            same = False        # the all-reduce goes through torch.distributed, which cannot be captured: eager every time
        if self.use_graph and same and getattr(V, "graph", None) is not None:
            V.graph.launch()
            z = V.z
        elif self.use_graph and same and getattr(V, "seen", 0) >= 1 and self._capture is None:
            g = ops.Graph()
            g.begin()
            V.z = emit()
            g.end()
            V.graph = g
            g.launch()
            z = V.z
        else:
            z = emit()
            V.inputs, V.seen, V.graph = (val_spec.data_ptr(), val_aux.data_ptr()), getattr(V, "seen", 0) + 1, None
        v = V.out.cpu().tolist()
        return z, {k: v[i] for k, i in LOSS_SLOTS.items() if k != "mi_accum"}

    def val_style_metrics(self):
        """``(W[nstyle], rho[pairs])`` of the styles of the last ``validate`` call (float64 numpy): what the
        reference's ``shapiro(x).statistic`` / ``spearmanr(a, b).correlation`` return for those columns."""
        return self._val_plan.metrics.read()
