"""Emitter for the 1-D-conv networks (``CompactEncoder`` / ``CompactDecoder``; reference
``sc/clustering/model.py:24-174, 264-295, 430-474``): the forward and the explicit backward of
every residual block as launches of the view-based HIP kernels in ``csrc/raae_conv.hip``.

A tensor is stored raw; PReLU, BatchNorm(affine=False, batch statistics) and the dropout scale
are applied by whoever reads it (``raae_view_t``), and BatchNorm's backward is folded into the
prologue of whoever back-propagates through it (``raae_grad_t``).  Block dataflow:

    R  = bn1(X)                          (view of the block input X; bn1 absent for 1 channel / length 1)
    T1 = conv1(R);  T2 = conv2(bn2(PReLU1(T1)))
    Sh = conv_short(R)                   (or the identity when shapes match)
    E1 = fc1(dropout(R)); E2 = fc2(PReLU(E1)); E3 = conv_excit(bn_excit(PReLU(E2)))   (if Cin != Cout)
    Y  = PReLU2(T2) + PReLUs(Sh) + PReLUe(E3 | E2)
"""
import os

import torch
from torch import nn

from . import ops
from ._lib import (IN_NONE, OUT_RAW, OUT_STATS_PRELU, OUT_STATS_RAW, OUT_SOFTPLUS, OUT_RELU, G_DIRECT,
                   RAAE_MAX_PARTS)


def _conv_desc(m, Lin):
    if isinstance(m, nn.ConvTranspose1d):
        k, s = m.kernel_size[0], m.stride[0]
        return ops.make_conv(m.in_channels, Lin, m.out_channels, Lin * s, k, s, 0, False, m.groups, True), Lin * s
    k, s, p = m.kernel_size[0], m.stride[0], m.padding[0]
    Lout = (Lin + 2 * p - k) // s + 1
    return ops.make_conv(m.in_channels, Lin, m.out_channels, Lout, k, s, p, m.padding_mode == "replicate",
                         m.groups, False), Lout


class Block:
    """Static description of one Encoding/DecodingBlock."""

    def __init__(self, m, Lin):
        self.m = m
        self.Cin, self.Cout, self.Lin = m.conv1.in_channels, m.conv1.out_channels, Lin
        self.cv1, self.L1 = _conv_desc(m.conv1, Lin)
        self.cv2, self.Lout = _conv_desc(m.conv2, self.L1)
        self.cvs = _conv_desc(m.conv_short, Lin)[0] if m.conv_short is not None else None
        self.E = m.fc1.out_features
        assert m.fc1.in_features == Lin and m.fc2.out_features == self.Lout
        self.cve = _conv_desc(m.conv_excit, self.Lout)[0] if m.conv_excit is not None else None
        self.p = m.dropout_1.p if m.dropout_1 is not None else 0.0
        self.bns = [b for b in (m.bn1, m.bn2, m.bn_excit) if b is not None]

    def params(self):
        return list(self.m.parameters())


class CompactNet:
    def __init__(self, module, kind, eng):
        self.module, self.kind, self.eng = module, kind, eng
        self.fused = bool(eng.cfg.get("fused_blocks", True))
        self.blocks = []
        if kind == "enc":
            L = module.lin3.in_features * 8      # dim_in = 256: three blocks 256 -> 64 -> 16 -> 8, 4 channels
            L = module.main[0].fc1.in_features
            for m in module.main:
                blk = Block(m, L)
                self.blocks.append(blk)
                L = blk.Lout
            self.flat = self.blocks[-1].Cout * L
            assert self.flat == module.lin3.in_features
            self.in_dim, self.out_dim = self.blocks[0].Lin, module.lin3.out_features
            self.bn_modules = [b for blk in self.blocks for b in blk.bns] + [module.bn_style]
        else:
            L = 1
            mods = list(module.main)
            for m in mods[:-3]:
                blk = Block(m, L)
                self.blocks.append(blk)
                L = blk.Lout
            self.bn_f, self.conv_f, act = mods[-3], mods[-2], mods[-1]
            self.cvf = _conv_desc(self.conv_f, L)[0]
            self.act = OUT_RELU if isinstance(act, nn.ReLU) else OUT_SOFTPLUS
            self.in_dim, self.out_dim = self.blocks[0].Cin, L
            self.bn_modules = [b for blk in self.blocks for b in blk.bns] + [self.bn_f]
        # forward_steps yields only at fused-block launches: with a per-layer first block it would run through to
        # its output before the paired chain has issued anything (StepEngine.emit_step, `pair`)
        self.pairable = self.fused and self.blocks[0].Cin <= 8 and self.blocks[0].Cout <= 8

    # ------------------------------------------------------------------ workspaces
    def alloc(self, b):
        dev = self.eng.device
        ws = type("WS", (), {})()
        ws.b = b

        def t(*shape):
            return torch.empty(*shape, device=dev)

        def parts(C):
            return torch.zeros(RAAE_MAX_PARTS, C, 2, dtype=torch.float64, device=dev)
        ws.blk = []
        for k in self.blocks:
            w = type("BW", (), {})()
            w.T1, w.T2 = t(b, k.Cout, k.L1), t(b, k.Cout, k.Lout)
            w.Sh = t(b, k.Cout, k.Lout) if k.cvs is not None else None
            w.E1, w.E2 = t(b, k.Cin, k.E), t(b, k.Cin, k.Lout)
            w.E3 = t(b, k.Cout, k.Lout) if k.cve is not None else None
            w.Y = t(b, k.Cout, k.Lout)
            # r*: the partial rows the producers write; p*: what the consumers read -- the same rows, or (experiment
            # `collapse_stats`) one row into which raae_stat_collapse2 has added them (c*)
            w.rT1, w.rE2, w.rY = parts(k.Cout), parts(k.Cin), parts(k.Cout)
            w.cT1, w.cE2, w.cY = (torch.zeros(1, C_, 2, dtype=torch.float64, device=dev) for C_ in (k.Cout, k.Cin, k.Cout))
            w.pT1, w.pE2, w.pY = w.rT1, w.rE2, w.rY
            w.nT1 = w.nE2 = w.nY = 0
            w.dBn2, w.pdBn2 = t(b, k.Cout, k.L1), parts(k.Cout)
            w.dR, w.pdR = t(b, k.Cin, k.Lin), parts(k.Cin)
            w.dBnE, w.pdBnE = t(b, k.Cin, k.Lout), parts(k.Cin)
            w.dE1a = t(b, k.Cin, k.E)
            w.dT2, w.dSh, w.dEx = t(b, k.Cout, k.Lout), t(b, k.Cout, k.Lout), t(b, k.Cout, k.Lout)
            w.dT1, w.dE2, w.dE1 = t(b, k.Cout, k.L1), t(b, k.Cin, k.Lout), t(b, k.Cin, k.E)
            w.ndBn2 = w.ndR = w.ndBnE = 0
            ws.blk.append(w)
        if self.kind == "enc":
            ws.zl = t(b, self.out_dim)
            ws.pzl = parts(self.out_dim)
            ws.nzl = 0
            ws.styles, ws.dz = t(b, self.out_dim), t(b, self.out_dim)
            ws.dYlast = t(b, self.flat)
            ws.out = ws.styles
        else:
            ws.spec = t(b, self.out_dim)
            ws.dBnF, ws.pdBnF = t(b, self.blocks[-1].Cout, self.out_dim), parts(self.blocks[-1].Cout)
            ws.ndBnF = 0
            ws.out = ws.spec
        return ws

    def mask_slots(self, tape, b, train=True):
        masks = []
        for k in self.blocks:
            if train and k.p > 0:
                off = tape.slot(b * k.Cin * k.Lin, 1, 1.0 - k.p)
                tape.draw("mask", off, (b, k.Cin, k.Lin), 1.0 - k.p)
                masks.append((off, (b, k.Cin, k.Lin)))
            else:
                masks.append(None)
        return masks

    # ------------------------------------------------------------------ helpers
    def _bn(self, bn, partials, nparts, count, train, update):
        if train:
            return ops.make_bn(partials, nparts, count, bn.running_mean, bn.running_var, bn.momentum, bn.eps, update)
        return ops.make_bn(None, 0, 0, bn.running_mean, bn.running_var, bn.momentum, bn.eps, False)

    def _mask(self, masks, i, train):
        if train and masks is not None and masks[i] is not None:
            return self.eng.tape.view(masks[i][0], *masks[i][1])
        return None

    def _cw(self, go, b, cv, view, conv, prelu, tag=0):
        """conv parameter gradients -> slabs; records the slab count of every tensor written."""
        eng = self.eng
        ps = [conv.weight, conv.bias] + ([prelu.weight] if prelu is not None else [])
        with eng.side_stream():
            ns = ops.conv_bwd_weight(go, b, cv, view, eng.gslab(conv.weight), eng.gslab(conv.bias),
                                     eng.gslab(prelu.weight) if prelu is not None else None, eng.arena.n)
        eng.note_slabs(ps, ns)

    def _lw(self, go, b, Cc, E, view, Lin, lin, prelu, tag=0):
        eng = self.eng
        with eng.side_stream():
            ns = ops.lenlin_bwd_weight(go, b, Cc, E, view, Lin, eng.gslab(lin.weight), eng.gslab(lin.bias),
                                       eng.gslab(prelu.weight) if prelu is not None else None, eng.arena.n)
        eng.note_slabs([lin.weight, lin.bias] + ([prelu.weight] if prelu is not None else []), ns)

    # ------------------------------------------------------------------ forward
    def forward(self, ws, x, masks, train=True):
        """One forward pass, every launch on its own."""
        steps = self.forward_steps(ws, x, masks, train)
        try:
            kind, args, nbytes = next(steps)
            while True:
                n = ops.block_fwd_a(args) if kind == "a" else ops.block_fwd_b(args)
                kind, args, nbytes = steps.send(n)
        except StopIteration as done:
            return done.value

    @staticmethod
    def forward_pair(first, second):
        """Two independent forward passes (``forward_steps`` generators, the ENCODER's first) in lockstep: while both
        are at the same phase of a fused block the two kernels share one launch (raae_block_fwd_a2 / _b2).
        Returns the two outputs."""
        gens, cur, out = [first, second], [None, None], [None, None]
        for j in (0, 1):
            try:
                cur[j] = next(gens[j])
            except StopIteration as done:
                out[j], gens[j] = done.value, None

        def advance(j, n):
            try:
                cur[j] = gens[j].send(n)
            except StopIteration as done:
                out[j], gens[j], cur[j] = done.value, None, None
        while gens[0] is not None or gens[1] is not None:
            if gens[0] is not None and gens[1] is not None and cur[0][0] == cur[1][0]:
                n1, n2 = ops.block_fwd_pair(cur[0][0], cur[0][1], cur[1][1])
                advance(0, n1)
                advance(1, n2)
                continue
            j = 0 if gens[0] is not None else 1
            kind, args, _ = cur[j]
            advance(j, ops.block_fwd_a(args) if kind == "a" else ops.block_fwd_b(args))
        return out[0], out[1]

    def forward_steps(self, ws, x, masks, train=True):
        """Generator form of the forward pass: yields ``(phase, args, algorithmic bytes)`` at every fused-block
        launch and expects the launch's partial-row count back; everything else is launched inline."""
        b = ws.b
        X, pX, nX = x, None, 0                # block input (raw), statistics of it
        for i, (k, w) in enumerate(zip(self.blocks, ws.blk)):
            m = k.m
            w.X, w.pX, w.nX = X, pX, nX

            def vR(update, mask=None, k=k, m=m, X=X, pX=pX, nX=nX):
                bn = self._bn(m.bn1, pX, nX, b * k.Lin, train, update) if m.bn1 is not None else None
                return ops.make_view(X, None, bn, mask)
            fused = self.fused and k.Cin <= 8 and k.Cout <= 8
            if fused:
                # two kernels per block: everything that only needs bn1, then everything that needs bn2 / bn_excit
                mask_i = self._mask(masks, i, train)
                n1 = yield ("a", ops.block_fwd_a_args(vR(True), mask_i, b, k, m, w.T1, w.Sh, w.E1, w.E2, w.rT1,
                                                      w.rE2 if k.cve is not None else None), 0)
                w.pT1, w.pE2, w.nT1, w.nE2 = w.rT1, w.rE2, n1, n1
                collapse = train and getattr(self, "collapse", False) and n1 >= self.eng.collapse_min_rows
                if collapse:
                    if k.cve is not None:
                        ops.stat_collapse2(w.rT1, n1, k.Cout, w.cT1, w.rE2, n1, k.Cin, w.cE2)
                        w.pE2, w.nE2 = w.cE2, 1
                    else:
                        ops.stat_collapse2(w.rT1, n1, k.Cout, w.cT1)
                    w.pT1, w.nT1 = w.cT1, 1
                v1 = ops.make_view(w.T1, m.relu1.weight, self._bn(m.bn2, w.pT1, w.nT1, b * k.L1, train, True))
                if k.cve is not None:
                    ve2 = ops.make_view(w.E2, m.relu_excit_2.weight,
                                        self._bn(m.bn_excit, w.pE2, w.nE2, b * k.Lout, train, True))
                else:
                    ve2 = ops.make_view(w.E2, m.relu_excit_2.weight)
                w.nY = yield ("b", ops.block_fwd_b_args(v1, ve2, vR(False) if k.cvs is None else None, b, k, m, w.Sh,
                                                        w.T2, w.E3, w.Y, w.rY), 0)
                w.pY = w.rY
                if collapse:
                    ops.stat_collapse2(w.rY, w.nY, k.Cout, w.cY)
                    w.pY, w.nY = w.cY, 1
                X, pX, nX = w.Y, w.pY, w.nY
                continue
            w.pT1, w.pE2, w.pY = w.rT1, w.rE2, w.rY          # (per-layer path: never collapsed)
            w.nT1 = ops.conv_fwd(vR(True), b, k.cv1, m.conv1.weight, m.conv1.bias, w.T1, OUT_STATS_PRELU,
                                 m.relu1.weight, w.pT1)
            v1 = ops.make_view(w.T1, m.relu1.weight, self._bn(m.bn2, w.pT1, w.nT1, b * k.L1, train, True))
            ops.conv_fwd(v1, b, k.cv2, m.conv2.weight, m.conv2.bias, w.T2, OUT_RAW)
            if k.cvs is not None:
                ops.conv_fwd(vR(False), b, k.cvs, m.conv_short.weight, m.conv_short.bias, w.Sh, OUT_RAW)
            ops.lenlin_fwd(vR(False, self._mask(masks, i, train)), b, k.Cin, k.Lin, m.fc1.weight, m.fc1.bias, k.E,
                           w.E1, OUT_RAW)
            ve1 = ops.make_view(w.E1, m.relu_excit_1.weight)
            if k.cve is not None:
                w.nE2 = ops.lenlin_fwd(ve1, b, k.Cin, k.E, m.fc2.weight, m.fc2.bias, k.Lout, w.E2, OUT_STATS_PRELU,
                                       m.relu_excit_2.weight, w.pE2)
                ve2 = ops.make_view(w.E2, m.relu_excit_2.weight,
                                    self._bn(m.bn_excit, w.pE2, w.nE2, b * k.Lout, train, True))
                ops.conv_fwd(ve2, b, k.cve, m.conv_excit.weight, m.conv_excit.bias, w.E3, OUT_RAW)
                vc = ops.make_view(w.E3, m.relu_excit_3.weight)
            else:
                ops.lenlin_fwd(ve1, b, k.Cin, k.E, m.fc2.weight, m.fc2.bias, k.Lout, w.E2, OUT_RAW)
                vc = ops.make_view(w.E2, m.relu_excit_2.weight)
            vb = ops.make_view(w.Sh, m.relu_short.weight) if k.cvs is not None else vR(False)
            w.nY = ops.sum3_fwd(ops.make_view(w.T2, m.relu2.weight), vb, vc, b, k.Cout, k.Lout, w.Y, w.pY)
            X, pX, nX = w.Y, w.pY, w.nY
        last = self.blocks[-1]
        if self.kind == "enc":
            mod = self.module
            ws.nzl = ops.dense_fwd(X.view(b, self.flat), b, self.flat, IN_NONE, None, None, None, mod.lin3.weight,
                                   mod.lin3.bias, self.out_dim, ws.zl, OUT_STATS_RAW, None, ws.pzl)
            ops.style_bn_fwd(ws.zl, b, self.out_dim, self._bn(mod.bn_style, ws.pzl, ws.nzl, b, train, True), ws.styles)
        else:
            vf = ops.make_view(X, None, self._bn(self.bn_f, pX, nX, b * last.Lout, train, True))
            ops.conv_fwd(vf, b, self.cvf, self.conv_f.weight, self.conv_f.bias, ws.spec.view(b, 1, self.out_dim),
                         OUT_RAW, None, None, self.act)
        if train:
            self.eng.count_bn(self.bn_modules)
        return ws.out

    # ------------------------------------------------------------------ backward
    def backward(self, ws, x, masks, g_out, dx_in=None, pending=None, keep_pending=False):
        """``pending``: weight-gradient tasks another network's backward left over (its return value with
        ``keep_pending``): they ride in this network's first fused launch instead of a launch of their own."""
        eng, b = self.eng, ws.b
        G = eng.gslab
        last, wl = self.blocks[-1], ws.blk[-1]
        if self.kind == "enc":
            mod = self.module
            ops.style_bn_bwd(g_out, ws.styles, b, self.out_dim, self._bn(mod.bn_style, ws.pzl, ws.nzl, b, True, False),
                             ws.dz)
            ns = ops.dense_bwd(ws.dz, G_DIRECT, None, 0, None, None, None, b, self.out_dim, wl.Y.view(b, self.flat),
                               self.flat, IN_NONE, None, None, None, mod.lin3.weight, G(mod.lin3.weight),
                               G(mod.lin3.bias), None, eng.arena.n, ws.dYlast, None)
            eng.note_slabs([mod.lin3.weight, mod.lin3.bias], ns)
            gy = dict(g=ws.dYlast.view(b, last.Cout, last.Lout), bn=None, parts=None, nparts=0)
        else:
            go = ops.make_grad(g_out.view(b, 1, self.out_dim), raw=ws.spec.view(b, 1, self.out_dim), act=self.act)
            vf = ops.make_view(wl.Y, None, self._bn(self.bn_f, wl.pY, wl.nY, b * last.Lout, True, False))
            if bool(eng.cfg.get("fused_head", True)) and ops.head_bwd_supported(go, b, self.cvf, vf):
                # one streaming pass: data gradient, its BatchNorm-backward sums and the head's parameter gradients
                ws.ndBnF, ns = ops.head_bwd(go, b, self.cvf, self.conv_f.weight, vf, ws.dBnF, ws.pdBnF,
                                            G(self.conv_f.weight), G(self.conv_f.bias), eng.arena.n)
                eng.note_slabs([self.conv_f.weight, self.conv_f.bias], ns)
            else:
                if pending is None and not eng._branch and self.fused and last.Cin <= 8 and last.Cout <= 8:
                    # serial chain: the head's weight gradient rides in the last block's backward phase B
                    head = [(go, self.cvf, vf, self.conv_f)]
                    pending = (ops.block_wgrad_args(b, [(go, self.cvf, vf, G(self.conv_f.weight), G(self.conv_f.bias))],
                                                    [], eng.arena.n), head, [])
                else:
                    self._cw(go, b, self.cvf, vf, self.conv_f, None)
                ws.ndBnF = ops.conv_bwd_data(go, b, self.cvf, self.conv_f.weight, vf, ws.dBnF, False, ws.pdBnF)
            gy = dict(g=ws.dBnF, bn=self.bn_f, parts=ws.pdBnF, nparts=ws.ndBnF)

        # Serial chain (batches below eng.overlap_min_batch): a block's weight-gradient tasks wait as `pending` and
        # ride in the launch of the NEXT block's backward phase B (raae_block_bwd_b_wgrad); the last ones go alone
        # or are handed to the network whose backward follows.

        def note_wgrad(pend, ns):
            _, convs_, lins_ = pend
            for (_, _, _, mod), n_ in zip(convs_, ns):
                eng.note_slabs([mod.weight, mod.bias], n_)
            for (_, _, _, _, _, mod), n_ in zip(lins_, ns[len(convs_):]):
                eng.note_slabs([mod.weight, mod.bias], n_)

        def flush(pend):
            if pend is not None:
                note_wgrad(pend, ops.block_wgrad(b, None, None, eng.arena.n, args=pend[0]))
            return None

        forked = []          # branched graph: blocks whose weight-gradient launches wait for the next fork

        def fork_wgrad():
            if forked:
                # at most TWO forked batches in flight: with main chain and auxiliary stream that is four concurrent
                # branches, the most the graph executor of ROCm 7.0 / 7.2 replays safely -- a captured step with five
                # (depth 3) crashed inside hipGraphLaunch in long sessions (rocgdb: hip::Graph::UpdateStreams reading a
                # stale hip::Stream*, or AllocCaptureSetValidate under GraphKernelNode::CreateCommand), alone it passed
                eng.join_side_streams(keep=min(int(eng.cfg.get("wgrad_overlap_depth", 2)), 2) - 1)
                with eng.side_stream():
                    for pend in forked:
                        note_wgrad(pend, ops.block_wgrad(b, None, None, eng.arena.n, args=pend[0]))
                forked.clear()

        for i in reversed(range(len(self.blocks))):
            k, w, m = self.blocks[i], ws.blk[i], self.blocks[i].m
            need_dx = i > 0 or dx_in is not None
            dR = w.dR if i > 0 else (dx_in.view(b, k.Cin, k.Lin) if dx_in is not None else None)
            ybn = self._bn(gy["bn"], w.pY, w.nY, b * k.Lout, True, False) if gy["bn"] is not None else None

            def gspec(raw, slope, gy=gy, ybn=ybn, w=w):
                return ops.make_grad(gy["g"], raw=raw, slope=slope, bn=ybn, g_partials=gy["parts"],
                                     g_nparts=gy["nparts"], u=w.Y if ybn is not None else None)

            def vR(mask=None, k=k, m=m, w=w):
                bn = self._bn(m.bn1, w.pX, w.nX, b * k.Lin, True, False) if m.bn1 is not None else None
                return ops.make_view(w.X, None, bn, mask)
            if self.fused and k.Cin <= 8 and k.Cout <= 8:
                bn2v = self._bn(m.bn2, w.pT1, w.nT1, b * k.L1, True, False)
                v1 = ops.make_view(w.T1, m.relu1.weight, bn2v)
                bne = self._bn(m.bn_excit, w.pE2, w.nE2, b * k.Lout, True, False) if k.cve is not None else None
                ve2 = ops.make_view(w.E2, m.relu_excit_2.weight, bne)
                if pending is not None:
                    nB, ns_w = ops.block_bwd_b(gspec(None, None), v1, ve2 if k.cve is not None else None, b, k, m, w,
                                               eng.arena.n, G, wgrad=pending[0])
                    note_wgrad(pending, ns_w)
                    pending = None
                else:
                    nB = ops.block_bwd_b(gspec(None, None), v1, ve2 if k.cve is not None else None, b, k, m, w,
                                         eng.arena.n, G)
                eng.note_slabs([m.relu2.weight] + ([m.relu_short.weight] if k.cvs is not None else []) +
                               [m.relu_excit_3.weight if k.cve is not None else m.relu_excit_2.weight], nB)
                g1 = ops.make_grad(w.dBn2, raw=w.T1, slope=m.relu1.weight, bn=bn2v, g_partials=w.pdBn2, g_nparts=nB)
                ge = ops.make_grad(w.dBnE, raw=w.E2, slope=m.relu_excit_2.weight, bn=bne, g_partials=w.pdBnE,
                                   g_nparts=nB) if k.cve is not None else None
                dE2 = w.dE2 if k.cve is not None else w.dEx
                mask = self._mask(masks, i, True)
                w.ndR = ops.block_bwd_a(g1, ge, vR(), mask, b, k, m, w, dE2, dR if need_dx else None,
                                        w.pdR if (need_dx and m.bn1 is not None) else None, eng.arena.n, G)
                eng.note_slabs([m.relu1.weight, m.relu_excit_1.weight] +
                               ([m.relu_excit_2.weight] if k.cve is not None else []), w.ndR)
                # weight gradients from the materialised gradients (side streams; no BatchNorm prologue).
                # At most ONE block's weight-gradient kernels are in flight: the previous block's are
                # joined first.  (Measured on ROCm 7.2: with deeper cross-block overlap a captured graph
                # stopped being bitwise equal to eager launches although its dependency edges were
                # complete -- tests/test_engine_gpu.py::test_graph_replay_is_bitwise_eager guards this.)
                G_ = eng.gslab
                convs = [(ops.make_grad(w.dT2), k.cv2, v1, m.conv2), (ops.make_grad(w.dT1), k.cv1, vR(), m.conv1)]
                if k.cve is not None:
                    convs.append((ops.make_grad(w.dEx), k.cve, ve2, m.conv_excit))
                if k.cvs is not None:
                    convs.append((ops.make_grad(w.dSh), k.cvs, vR(), m.conv_short))
                lins = [(ops.make_grad(dE2), k.Cin, k.Lout, k.E, ops.make_view(w.E1, m.relu_excit_1.weight), m.fc2),
                        (ops.make_grad(w.dE1), k.Cin, k.E, k.Lin, vR(mask), m.fc1)]
                wargs = ops.block_wgrad_args(
                    b, [(g_, cv_, v_, G_(mod.weight), G_(mod.bias)) for g_, cv_, v_, mod in convs],
                    [(g_, c_, e_, l_, v_, G_(mod.weight), G_(mod.bias)) for g_, c_, e_, l_, v_, mod in lins],
                    eng.arena.n)
                if eng._branch:
                    # One fork of the captured graph carries the weight-gradient launches of `wgrad_fork_blocks`
                    # consecutive blocks (default 2): an edge between two hardware queues costs the main chain ~13 us
                    # (rocprofv3 timeline of the 4096-row step: every fork delayed the next backward kernel by that
                    # much), a weight-gradient launch that starts one block later costs nothing.
                    # `wgrad_overlap_depth` such batches may be in flight at once (default 2 since the end of round 3:
                    # 283.0 against 279.4 steps/s at 4096 rows, where in the decoder-last phases the side branch,
                    # not the main chain, ends the phase; 1: the previous batch is joined before the next forks).
                    forked.append((wargs, convs, lins))
                    # The network whose backward ENDS the phase (nothing upstream wants its input gradient): what is
                    # forked after its last block cannot hide behind anything -- the optimizer waits for it across a
                    # fork and a join edge (~13 + ~10 us on the 4096-row timeline, plus the launches themselves: 65-75 us
                    # of exposed tail per phase).  There the second-to-last block forks what is pending (it runs
                    # beside the last block's two backward kernels) and the last block's tasks stay on the main chain.
                    tail = dx_in is None and pending is None and bool(eng.cfg.get("wgrad_tail_inline", os.environ.get("RAAE_WGRAD_TAIL", "1") != "0"))
                    if tail and i == 0:
                        for pend in forked:
                            note_wgrad(pend, ops.block_wgrad(b, None, None, eng.arena.n, args=pend[0]))
                        forked.clear()
                    elif len(forked) >= int(eng.cfg.get("wgrad_fork_blocks", 2)) or i == 0 or (tail and i == 1):
                        fork_wgrad()
                else:
                    pending = (wargs, convs, lins)
                if i > 0:
                    gy = dict(g=w.dR, bn=m.bn1, parts=w.pdR, nparts=w.ndR)
                    if m.bn1 is None:
                        gy = dict(g=w.dR, bn=None, parts=None, nparts=0)
                continue
            # ---- main branch
            pending = flush(pending)
            go2 = gspec(w.T2, m.relu2.weight)
            v1 = ops.make_view(w.T1, m.relu1.weight, self._bn(m.bn2, w.pT1, w.nT1, b * k.L1, True, False))
            self._cw(go2, b, k.cv2, v1, m.conv2, m.relu2)
            w.ndBn2 = ops.conv_bwd_data(go2, b, k.cv2, m.conv2.weight, v1, w.dBn2, False, w.pdBn2)
            go1 = ops.make_grad(w.dBn2, raw=w.T1, slope=m.relu1.weight,
                                bn=self._bn(m.bn2, w.pT1, w.nT1, b * k.L1, True, False), g_partials=w.pdBn2,
                                g_nparts=w.ndBn2)
            self._cw(go1, b, k.cv1, vR(), m.conv1, m.relu1)
            if need_dx:
                ops.conv_bwd_data(go1, b, k.cv1, m.conv1.weight, vR(), dR, False, None)
            # ---- shortcut
            if k.cvs is not None:
                gos = gspec(w.Sh, m.relu_short.weight)
                self._cw(gos, b, k.cvs, vR(), m.conv_short, m.relu_short)
                if need_dx:
                    ops.conv_bwd_data(gos, b, k.cvs, m.conv_short.weight, vR(), dR, True, None)
            elif need_dx:
                ops.grad_materialize(gspec(None, None), b, k.Cout, k.Lout, dR, True, None, 0)
            # ---- excitation branch
            ve1 = ops.make_view(w.E1, m.relu_excit_1.weight)
            if k.cve is not None:
                goe3 = gspec(w.E3, m.relu_excit_3.weight)
                bne = self._bn(m.bn_excit, w.pE2, w.nE2, b * k.Lout, True, False)
                ve2 = ops.make_view(w.E2, m.relu_excit_2.weight, bne)
                self._cw(goe3, b, k.cve, ve2, m.conv_excit, m.relu_excit_3)
                w.ndBnE = ops.conv_bwd_data(goe3, b, k.cve, m.conv_excit.weight, ve2, w.dBnE, False, w.pdBnE)
                goe2 = ops.make_grad(w.dBnE, raw=w.E2, slope=m.relu_excit_2.weight, bn=bne, g_partials=w.pdBnE,
                                     g_nparts=w.ndBnE)
            else:
                goe2 = gspec(w.E2, m.relu_excit_2.weight)
            self._lw(goe2, b, k.Cin, k.Lout, ve1, k.E, m.fc2, m.relu_excit_2)
            ops.lenlin_bwd_data(goe2, b, k.Cin, k.Lout, m.fc2.weight, ve1, k.E, w.dE1a, False, None)
            goe1 = ops.make_grad(w.dE1a, raw=w.E1, slope=m.relu_excit_1.weight)
            vin = vR(self._mask(masks, i, True))
            self._lw(goe1, b, k.Cin, k.E, vin, k.Lin, m.fc1, m.relu_excit_1)
            if need_dx:
                w.ndR = ops.lenlin_bwd_data(goe1, b, k.Cin, k.E, m.fc1.weight, vin, k.Lin, dR, True,
                                            w.pdR if m.bn1 is not None else None)
            if i > 0:
                gy = dict(g=w.dR, bn=m.bn1, parts=w.pdR, nparts=w.ndR)
                if m.bn1 is None:
                    gy = dict(g=w.dR, bn=None, parts=None, nparts=0)
        fork_wgrad()
        if keep_pending:
            return pending
        flush(pending)
        return None
