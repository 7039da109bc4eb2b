"""Per-kernel parity tests (GPU): each C-ABI entry point against a plain PyTorch fp32
CPU reference of the same op (autograd for the backward).  Tolerances are written next
to each assert; fp32 everywhere, differences come from summation order only.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from rankaae_amd import ops, _lib
    DEV = torch.device("cuda:0")


def dev(t, dtype=torch.float32):
    return t.to(dtype).contiguous().to(DEV)


def close(a, b, rtol, atol, what=""):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.all(err <= tol), f"{what}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}, " \
                               f"ref {b.flat[err.argmax()]:.6e} got {a.flat[err.argmax()]:.6e}"


def _chain_ref(x, w0, b0, s0, mask0, w1, b1, out_kind):
    z0 = F.linear(x, w0, b0)
    a0 = F.prelu(z0, s0)
    y0 = F.batch_norm(a0, None, None, training=True, eps=1e-5)
    x1 = y0 * mask0 if mask0 is not None else y0
    z1 = F.linear(x1, w1, b1)
    if out_kind == "softplus":
        return z0, F.softplus(z1, beta=2), y0
    return z0, z1, y0


@pytest.mark.parametrize("B,K,H,N,outk,drop", [
    (256, 256, 64, 64, "raw", True), (36, 256, 64, 6, "raw", True), (100, 6, 64, 256, "softplus", True),
    (64, 13, 64, 512, "softplus", False), (77, 512, 64, 13, "raw", True), (4096, 64, 64, 64, "raw", True),
    (5000, 256, 64, 6, "raw", True),
])
def test_dense_chain_fwd_bwd(B, K, H, N, outk, drop):
    g = torch.Generator().manual_seed(B * 7 + K)
    x = torch.randn(B, K, generator=g)
    w0 = torch.randn(H, K, generator=g) / K ** 0.5
    b0 = torch.randn(H, generator=g) * 0.1
    s0 = torch.rand(H, generator=g) * 0.5 - 0.1          # PReLU slopes, some negative
    w1 = torch.randn(N, H, generator=g) / H ** 0.5
    b1 = torch.randn(N, generator=g) * 0.1
    mask0 = (torch.rand(B, H, generator=g) < 0.9).float() / 0.9 if drop else None
    gout = torch.randn(B, N, generator=g)

    # ---- reference (CPU autograd) ----
    xr = x.clone().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) for t in (w0, b0, s0, w1, b1)]
    z0r, outr, y0r = _chain_ref(xr, pr[0], pr[1], pr[2], mask0, pr[3], pr[4], outk)
    (outr * gout).sum().backward()

    # ---- HIP ----
    xd, w0d, b0d, s0d, w1d, b1d = map(dev, (x, w0, b0, s0, w1, b1))
    md = dev(mask0) if drop else None
    z0 = torch.empty(B, H, device=DEV)
    out = torch.empty(B, N, device=DEV)
    part0 = torch.zeros(_lib.RAAE_MAX_PARTS, H, 2, dtype=torch.float64, device=DEV)
    np0 = ops.dense_fwd(xd, B, K, _lib.IN_NONE, None, None, None, w0d, b0d, H, z0, _lib.OUT_STATS_PRELU, s0d, part0)
    rm = torch.zeros(H, device=DEV)
    rv = torch.ones(H, device=DEV)
    bn0 = ops.make_bn(part0, np0, B, rm, rv, update_running=True)
    ok = _lib.OUT_SOFTPLUS if outk == "softplus" else _lib.OUT_RAW
    ops.dense_fwd(z0, B, H, _lib.IN_PRELU_BN_DROP, s0d, bn0, md, w1d, b1d, N, out, ok)
    torch.cuda.synchronize()
    close(z0, z0r, 2e-5, 2e-5, "z0")
    close(out, outr, 1e-4, 1e-4, "out")
    # running stats as torch.nn.BatchNorm1d would update them (momentum 0.1, unbiased var)
    a0 = F.prelu(z0r.detach(), s0)
    close(rm, 0.1 * a0.mean(0), 1e-4, 1e-6, "running_mean")
    close(rv, 0.9 + 0.1 * a0.var(0, unbiased=True), 1e-4, 1e-6, "running_var")

    # backward: layer 1 then layer 0
    bn0.update_running = 0
    n_all = H * K + H + H + N * H + N          # slab layout: w0 | b0 | s0 | w1 | b1
    off_w0, off_b0, off_s0, off_w1, off_b1 = 0, H * K, H * K + H, H * K + 2 * H, H * K + 2 * H + N * H
    stride = (n_all + 63) // 64 * 64
    slabs = torch.zeros(256, stride, device=DEV)
    gd = dev(gout)
    dx1 = torch.empty(B, H, device=DEV)
    dxp1 = torch.zeros(_lib.RAAE_MAX_PARTS, H, 2, dtype=torch.float64, device=DEV)
    gk = _lib.G_SOFTPLUS if outk == "softplus" else _lib.G_DIRECT
    ns1 = ops.dense_bwd(gd, gk, None, 0, out, None, None, B, N, z0, H, _lib.IN_PRELU_BN_DROP, s0d, bn0, md, w1d,
                        slabs[0, off_w1:], slabs[0, off_b1:], None, stride, dx1, dxp1)
    dx0 = torch.empty(B, K, device=DEV)
    ns0 = ops.dense_bwd(dx1, _lib.G_PRELU_BN, dxp1, ns1, z0, s0d, bn0, B, H, xd, K, _lib.IN_NONE, None, None, None, w0d,
                        slabs[0, off_w0:], slabs[0, off_b0:], slabs[0, off_s0:], stride, dx0, None)
    torch.cuda.synchronize()
    scale = float(gout.abs().mean()) * B ** 0.5
    dw1 = slabs[:ns1, off_w1:off_w1 + N * H].sum(0).view(N, H)
    db1 = slabs[:ns1, off_b1:off_b1 + N].sum(0)
    dw0 = slabs[:ns0, off_w0:off_w0 + H * K].sum(0).view(H, K)
    db0 = slabs[:ns0, off_b0:off_b0 + H].sum(0)
    ds0 = slabs[:ns0, off_s0:off_s0 + H].sum(0)
    close(dw1, pr[3].grad, 2e-4, 2e-5 * scale, "dw1")
    close(db1, pr[4].grad, 2e-4, 2e-5 * scale, "db1")
    close(dw0, pr[0].grad, 5e-4, 5e-5 * scale, "dw0")
    close(db0, pr[1].grad, 5e-4, 5e-5 * scale, "db0")
    close(ds0, pr[2].grad, 5e-4, 5e-5 * scale, "dslope0")
    close(dx0, xr.grad, 5e-4, 5e-5, "dx0")


def test_dense_prelu_drop_no_bn():
    """Discriminator-style layers: PReLU -> Dropout -> Linear, no BatchNorm."""
    g = torch.Generator().manual_seed(5)
    B, K, H = 300, 6, 64
    x = torch.randn(B, K, generator=g)
    w0, b0, s0 = torch.randn(H, K, generator=g) * 0.4, torch.randn(H, generator=g) * 0.1, torch.rand(H, generator=g) * 0.3
    w1, b1 = torch.randn(1, H, generator=g) * 0.2, torch.randn(1, generator=g)
    mask = (torch.rand(B, H, generator=g) < 0.94).float() / 0.94
    gout = torch.randn(B, 1, generator=g)
    xr = x.clone().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) for t in (w0, b0, s0, w1, b1)]
    z0r = F.linear(xr, pr[0], pr[1])
    outr = F.linear(F.prelu(z0r, pr[2]) * mask, pr[3], pr[4])
    (outr * gout).sum().backward()
    xd, w0d, b0d, s0d, w1d, b1d, md, gd = map(dev, (x, w0, b0, s0, w1, b1, mask, gout))
    z0 = torch.empty(B, H, device=DEV)
    out = torch.empty(B, 1, device=DEV)
    ops.dense_fwd(xd, B, K, _lib.IN_NONE, None, None, None, w0d, b0d, H, z0, _lib.OUT_RAW)
    ops.dense_fwd(z0, B, H, _lib.IN_PRELU_DROP, s0d, None, md, w1d, b1d, 1, out, _lib.OUT_RAW)
    close(out, outr, 1e-5, 1e-5, "logit")
    stride = 1024
    slabs = torch.zeros(256, stride, device=DEV)
    dx1 = torch.empty(B, H, device=DEV)
    ns1 = ops.dense_bwd(gd, _lib.G_DIRECT, None, 0, None, None, None, B, 1, z0, H, _lib.IN_PRELU_DROP, s0d, None, md,
                        w1d, slabs[0, 0:], slabs[0, 64:], None, stride, dx1, None)
    dx0 = torch.empty(B, K, device=DEV)
    ns0 = ops.dense_bwd(dx1, _lib.G_PRELU, None, 0, z0, s0d, None, B, H, xd, K, _lib.IN_NONE, None, None, None, w0d,
                        slabs[0, 128:], slabs[0, 512:], slabs[0, 576:], stride, dx0, None)
    close(slabs[:ns1, 0:H].sum(0).view(1, H), pr[3].grad, 1e-4, 1e-4, "dw1")
    close(slabs[:ns1, 64:65].sum(0), pr[4].grad, 1e-4, 1e-4, "db1")
    close(slabs[:ns0, 128:128 + H * K].sum(0).view(H, K), pr[0].grad, 1e-4, 1e-4, "dw0")
    close(slabs[:ns0, 512:512 + H].sum(0), pr[1].grad, 1e-4, 1e-4, "db0")
    close(slabs[:ns0, 576:576 + H].sum(0), pr[2].grad, 1e-4, 1e-4, "ds0")
    close(dx0, xr.grad, 1e-4, 1e-5, "dx0")


@pytest.mark.parametrize("B,C", [(256, 6), (36, 13), (4096, 6), (1050, 64)])
def test_style_bn(B, C):
    g = torch.Generator().manual_seed(B + C)
    z = torch.randn(B, C, generator=g) * 2 + 0.5
    gout = torch.randn(B, C, generator=g)
    zr = z.clone().requires_grad_(True)
    yr = F.batch_norm(zr, None, None, training=True, eps=1e-5)
    (yr * gout).sum().backward()
    # partials as a producer would emit them (two "workgroups")
    part = torch.zeros(4, C, 2, dtype=torch.float64)
    h = B // 2
    for i, sl in enumerate((z[:h], z[h:])):
        part[i, :, 0] = sl.double().sum(0)
        part[i, :, 1] = (sl.double() ** 2).sum(0)
    pd = part.to(DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    bn = ops.make_bn(pd, 2, B, rm, rv, update_running=True)
    y = torch.empty(B, C, device=DEV)
    ops.style_bn_fwd(dev(z), B, C, bn, y)
    close(y, yr, 1e-5, 1e-5, "styles")
    close(rm, 0.1 * z.mean(0), 1e-5, 1e-6, "rm")
    bn.update_running = 0
    dz = torch.empty(B, C, device=DEV)
    ops.style_bn_bwd(dev(gout), y, B, C, bn, dz)
    close(dz, zr.grad, 1e-4, 2e-5, "dz")
    # eval mode
    bn_e = ops.make_bn(None, 0, 0, rm, rv)
    ops.style_bn_fwd(dev(z), B, C, bn_e, y)
    close(y, (z - rm.cpu()) / torch.sqrt(rv.cpu() + 1e-5), 1e-5, 1e-6, "eval")


@pytest.mark.parametrize("B,K,ld,act", [(256, 5, 6, True), (36, 5, 6, True), (1050, 5, 6, True), (300, 12, 13, True),
                                        (257, 1, 1, False), (4096, 5, 6, True)])
def test_rank_loss(B, K, ld, act):
    from oracle.ref_train import kendall_constraint, kendall_closed_form
    g = torch.Generator().manual_seed(B + K)
    d = torch.randn(B, K, generator=g)
    d[:, min(1, K - 1)] = torch.randint(4, 7, (B,), generator=g).float()     # ties
    z = torch.randn(B, ld, generator=g)
    if B <= 1100:
        zr = z.clone().requires_grad_(True)
        lr = kendall_constraint(d, zr[:, :K], activate=act)
        lr.backward()
        gref, lref = zr.grad, float(lr)
    else:   # literal form needs B^2*K floats; use the float64 closed form (pinned to it on CPU)
        l64, g64 = kendall_closed_form(d, z[:, :K], activate=act)
        gref = torch.zeros(B, ld, dtype=torch.float64)
        gref[:, :K] = g64
        lref = float(l64)
    work = torch.empty(ops.rank_loss_work_bytes(B, K), dtype=torch.uint8, device=DEV)
    loss = torch.zeros(1, device=DEV)
    dz = torch.full((B, ld), 7.0, device=DEV)
    ops.rank_loss_fwd_bwd(dev(d), K, dev(z), ld, B, K, act, work, loss, dz)
    assert abs(float(loss) - lref) <= 1e-5 * abs(lref) + 1e-7, (float(loss), lref)
    close(dz, gref, 1e-5, 1e-9, "dz")
    ops.rank_loss_fwd_bwd(dev(d), K, dev(z), ld, B, K, act, work, loss, None)   # validation form
    assert abs(float(loss) - lref) <= 1e-5 * abs(lref) + 1e-7


@pytest.mark.parametrize("B,L,scale", [(256, 256, True), (36, 256, True), (1050, 256, False), (64, 512, True)])
def test_recon_loss(B, L, scale):
    from oracle.ref_train import recon_loss
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, L, generator=g) + 0.2
    y = (x + 0.3 * torch.randn(B, L, generator=g)).requires_grad_(True)
    with torch.no_grad():
        y[0] *= 2.0      # clamp branch (ratio > 1.3)
        y[1] *= 0.3      # ratio < 0.7
    lr = recon_loss(x, y, scale=scale)
    lr.backward()
    part = torch.zeros(_lib.RAAE_MAX_PARTS, dtype=torch.float64, device=DEV)
    dy = torch.empty(B, L, device=DEV)
    n = ops.recon_loss_fwd_bwd(dev(x), dev(y.detach()), B, L, scale, part, dy)
    out = torch.zeros(8, device=DEV)
    ops.loss_finalize(part, n, 1.0, out, 2)
    assert abs(float(out[2]) - float(lr)) <= 1e-5 * abs(float(lr)), (float(out[2]), float(lr))
    close(dy, y.grad, 1e-4, 1e-9, "dy")


@pytest.mark.parametrize("B,L", [(256, 256), (36, 256), (64, 512), (5, 20)])
def test_smooth_loss(B, L):
    from oracle.ref_train import smoothness_loss
    from oracle.ref_model import gaussian_taps
    g = torch.Generator().manual_seed(L)
    x = (torch.rand(B, L, generator=g) + 0.1 * torch.randn(B, L, generator=g)).requires_grad_(True)
    lr = smoothness_loss(x, 17)
    lr.backward()
    part = torch.zeros(_lib.RAAE_MAX_PARTS, dtype=torch.float64, device=DEV)
    dx = torch.empty(B, L, device=DEV)
    n = ops.smooth_loss_fwd_bwd(dev(x.detach()), B, L, gaussian_taps(17, 3.0).tolist(), part, dx)
    out = torch.zeros(8, device=DEV)
    ops.loss_finalize(part, n, 1.0, out, 4)
    assert abs(float(out[4]) - float(lr)) <= 2e-5 * abs(float(lr)), (float(out[4]), float(lr))
    close(dx, x.grad, 1e-4, 1e-9, "dx")


def test_mse_and_bce():
    g = torch.Generator().manual_seed(1)
    a = torch.randn(300, 6, generator=g).requires_grad_(True)
    b = torch.randn(300, 6, generator=g)
    lr = F.mse_loss(a, b)
    lr.backward()
    part = torch.zeros(_lib.RAAE_MAX_PARTS, dtype=torch.float64, device=DEV)
    da = torch.empty(300, 6, device=DEV)
    n = ops.mse_fwd_bwd(dev(a.detach()), dev(b), 1800, part, da)
    out = torch.zeros(8, device=DEV)
    ops.loss_finalize(part, n, 1.0, out, 3, 5)
    ops.loss_finalize(part, n, 1.0, out, 3, 5)
    assert abs(float(out[3]) - float(lr)) < 1e-6 * float(lr)
    assert abs(float(out[5]) - 2 * float(lr)) < 1e-6 * float(lr)      # accumulating slot
    close(da, a.grad, 1e-5, 1e-9, "da")

    o = (torch.randn(256 + 36, generator=g) * 3).requires_grad_(True)
    bce = torch.nn.BCEWithLogitsLoss()
    lr = bce(o[:256], torch.ones(256)) + bce(o[256:], torch.zeros(36))
    lr.backward()
    loss = torch.zeros(1, device=DEV)
    do = torch.empty(292, device=DEV)
    ops.bce_pair_fwd_bwd(dev(o.detach()), 256, 36, loss, do)
    assert abs(float(loss) - float(lr)) < 1e-6 * float(lr)
    close(do, o.grad, 1e-5, 1e-9, "dlogits")


@pytest.mark.parametrize("decoupled,wd", [(True, 0.01), (False, 0.0), (False, 0.01)])
def test_adam_matches_torch(decoupled, wd):
    g = torch.Generator().manual_seed(11)
    n = 64 * 40
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    cls = torch.optim.AdamW if decoupled else torch.optim.Adam
    opt = cls([ref], lr=0.01, betas=(0.99, 0.9999), weight_decay=wd)
    p, m, v = dev(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    hyper = torch.tensor([0.01, 0.99, 0.9999, 1e-8, wd], dtype=torch.float64, device=DEV)
    step = torch.zeros(4, dtype=torch.int32, device=DEV)
    rngc = torch.zeros(3, dtype=torch.int64, device=DEV)          # {step counter, seed, the step's dropout hash keys}
    cursor = torch.zeros(1, dtype=torch.int32, device=DEV)
    nslab = 3
    seg = torch.full((n // 64,), nslab, dtype=torch.int16, device=DEV)
    seg[5] = 0          # one segment "without gradient": must be left untouched
    for it in range(5):
        slabs = torch.randn(nslab, n, generator=g) * (0.0 if it == 3 else 1.0)     # one all-zero gradient step
        grad = (slabs[0] + slabs[1]) + slabs[2]
        ref.grad = grad.clone()
        opt.step()
        ops.step_tick(step, 4, 0b0101, rngc, cursor, 256)
        ops.adam_step(p, m, v, dev(slabs), n, seg, n, hyper, step[2:], decoupled, max_nslab=(3 if it % 2 else 40))
    torch.cuda.synchronize()
    assert step.tolist() == [5, 0, 5, 0] and int(rngc[0]) == 5 and int(rngc[2]) != 0 and int(cursor) == 5 * 256
    pr = ref.detach().clone()
    pr[5 * 64:6 * 64] = p0[5 * 64:6 * 64]
    close(p, pr, 2e-6, 2e-7, "params after 5 steps")


def test_adam_many_slabs_lane_split():
    """Ranges with many gradient slabs take the 8-lanes-per-element kernel; same result as a host sum."""
    g = torch.Generator().manual_seed(5)
    n, rows = 64 * 6, 200
    slabs = torch.randn(rows, n, generator=g)
    counts = [200, 37, 0, 8, 65, 1]
    seg = torch.tensor(counts, dtype=torch.int16, device=DEV)
    p0 = torch.randn(n, generator=g)
    hyper = torch.tensor([0.01, 0.9, 0.999, 1e-8, 0.01], dtype=torch.float64, device=DEV)
    step = torch.ones(1, dtype=torch.int32, device=DEV)
    out = []
    for hint in (8, 200):                       # per-thread kernel, lane-split kernel
        p, m, v = dev(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        ops.adam_step(p, m, v, dev(slabs), n, seg, n, hyper, step, True, max_nslab=hint)
        out.append((p.cpu(), m.cpu(), v.cpu()))
    grad = torch.stack([slabs[:c, 64 * i:64 * (i + 1)].double().sum(0) for i, c in enumerate(counts)]).reshape(-1)
    close(out[1][1], 0.1 * grad.float(), 1e-5, 1e-6, "m = (1-beta1) * sum of the segment's slabs")
    for a, b in zip(out[0], out[1]):
        close(a, b, 1e-5, 1e-6, "lane-split vs per-thread")
    assert torch.equal(out[1][0][128:192], p0[128:192]), "segment without slabs untouched"


def test_rng_fill_statistics():
    n_norm, n_mask = 1 << 20, 1 << 20
    desc = torch.tensor([[0, n_norm, 0, 0], [n_norm, n_mask, 1, 0]], dtype=torch.int32, device=DEV)
    scale = torch.tensor([1.0, 0.96], device=DEV)
    tape = torch.zeros(n_norm + n_mask, device=DEV)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.rng_fill(tape, desc, scale, 2, n_norm + n_mask, 1234, ctr)
    a = tape[:n_norm].cpu().double()
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a ** 4).mean()) - 3.0) < 0.1
    mk = tape[n_norm:].cpu()
    vals = torch.unique(mk)
    assert len(vals) == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / 0.96) < 1e-6
    assert abs(float((mk > 0).float().mean()) - 0.96) < 2e-3
    t2 = torch.zeros_like(tape)
    ops.rng_fill(t2, desc, scale, 2, n_norm + n_mask, 1234, ctr)
    assert torch.equal(tape, t2)                       # same (seed, counter) -> same tape
    ctr += 1
    ops.rng_fill(t2, desc, scale, 2, n_norm + n_mask, 1234, ctr)
    assert not torch.equal(tape, t2)


def test_gather_batch_and_disc_input():
    g = torch.Generator().manual_seed(2)
    spec, aux = torch.randn(100, 32, generator=g), torch.randn(100, 5, generator=g)
    idx = torch.randperm(100, generator=g)
    noise = torch.randn(16, 32, generator=g)
    cursor = torch.tensor([48], dtype=torch.int32, device=DEV)
    so, ao = torch.empty(16, 32, device=DEV), torch.empty(16, 5, device=DEV)
    ops.gather_batch(dev(spec), dev(aux), idx.to(DEV), cursor, dev(noise), 0.02, 16, 32, 5, so, ao)
    rows = idx[32:48]
    close(so, spec[rows] + noise * 0.02, 1e-6, 1e-7, "spec")
    close(ao, aux[rows], 0, 0, "aux")
    zr, st, nz = torch.randn(8, 6, generator=g), torch.randn(5, 6, generator=g), torch.randn(13, 6, generator=g)
    out = torch.empty(13, 6, device=DEV)
    ops.disc_input(dev(zr), dev(st), dev(nz), 0.56, 8, 5, 6, out)
    close(out, torch.cat([zr, st]) + 0.56 * nz, 1e-6, 1e-7, "disc_input")
    al = torch.tensor([0.25], device=DEV)
    dst = torch.empty(5, 6, device=DEV)
    ops.scale_by_dev(out[8:], al, -1.0, 30, dst)
    close(dst, -0.25 * out[8:].cpu(), 1e-6, 0, "grl")


# ------------------------------------------------------------------ conv-network kernels
def _sandwich(op_kind, B, Cin, Lin, cfg, seed):
    """value = mask * BN(PReLU(X)) -> op -> out ; a = PReLU(out) ; y = BN(a) ; loss = sum(y * G)."""
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(B, Cin, Lin, generator=g)
    sX = torch.rand(Cin, generator=g) * 0.4 - 0.1
    mask = (torch.rand(B, Cin, Lin, generator=g) < 0.9).float() / 0.9
    if op_kind == "conv":
        m = torch.nn.Conv1d(Cin, cfg["Cout"], cfg["K"], stride=cfg["s"], padding=cfg["p"],
                            padding_mode="replicate" if cfg["rep"] else "zeros", groups=cfg["g"])
    elif op_kind == "convT":
        m = torch.nn.ConvTranspose1d(Cin, cfg["Cout"], cfg["K"], stride=cfg["K"], groups=cfg["g"])
    else:
        m = torch.nn.Linear(Lin, cfg["E"])
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    Xr = X.clone().requires_grad_(True)
    sXr = sX.clone().requires_grad_(True)
    v = F.batch_norm(F.prelu(Xr, sXr), None, None, training=True, eps=1e-5)
    v.retain_grad()
    out = m(v * mask)
    Cout = out.shape[1]
    sO = torch.rand(Cout, generator=g) * 0.4 - 0.1
    sOr = sO.clone().requires_grad_(True)
    y = F.batch_norm(F.prelu(out, sOr), None, None, training=True, eps=1e-5)
    G = torch.randn(y.shape, generator=g)
    (y * G).sum().backward()
    return dict(X=X, sX=sX, mask=mask, m=m, v=v, out=out, sO=sO, y=y, G=G, dsO=sOr.grad, dv=v.grad)


def _partials_of(t):      # [B,C,L] -> [2, C, 2] doubles split in two "workgroups" along B
    h = max(t.shape[0] // 2, 1)
    P = torch.zeros(2, t.shape[1], 2, dtype=torch.float64)
    for i, sl in enumerate((t[:h], t[h:])):
        P[i, :, 0] = sl.double().sum((0, 2))
        P[i, :, 1] = (sl.double() ** 2).sum((0, 2))
    return P


CONV_CASES = [
    ("conv", 64, 1, 256, dict(Cout=4, K=11, s=2, p=5, rep=True, g=1)),
    ("conv", 37, 4, 128, dict(Cout=4, K=11, s=2, p=5, rep=False, g=1)),
    ("conv", 16, 4, 64, dict(Cout=4, K=4, s=4, p=0, rep=False, g=4)),
    ("conv", 16, 4, 16, dict(Cout=4, K=5, s=1, p=2, rep=True, g=1)),
    ("conv", 16, 8, 64, dict(Cout=4, K=1, s=1, p=0, rep=False, g=4)),
    ("conv", 9, 6, 8, dict(Cout=8, K=1, s=1, p=0, rep=False, g=2)),
    ("conv", 8, 4, 256, dict(Cout=4, K=11, s=1, p=5, rep=True, g=1)),
    ("conv", 8, 4, 256, dict(Cout=1, K=1, s=1, p=0, rep=False, g=1)),
    # >= 2^20 outputs: the register-blocked large-batch forward (raae_conv_strip.inc), ragged last group
    ("conv", 1027, 4, 256, dict(Cout=4, K=11, s=1, p=5, rep=True, g=1)),
    ("conv", 2050, 4, 256, dict(Cout=4, K=11, s=2, p=5, rep=False, g=1)),
    ("conv", 2049, 1, 256, dict(Cout=4, K=11, s=2, p=5, rep=True, g=1)),
    ("conv", 8200, 4, 64, dict(Cout=4, K=7, s=2, p=3, rep=True, g=1)),
    ("conv", 4100, 4, 64, dict(Cout=4, K=5, s=1, p=2, rep=False, g=1)),
    ("convT", 16, 6, 1, dict(Cout=8, K=2, g=1)),
    ("convT", 16, 8, 2, dict(Cout=8, K=4, g=1)),
    ("convT", 16, 6, 1, dict(Cout=8, K=8, g=2)),
    ("convT", 16, 8, 8, dict(Cout=4, K=8, g=4)),
    ("convT", 33, 4, 64, dict(Cout=4, K=2, g=1)),
    ("lenlin", 40, 4, 64, dict(E=2)),
    ("lenlin", 40, 1, 256, dict(E=4)),
    ("lenlin", 40, 8, 2, dict(E=64)),
    ("lenlin", 40, 6, 1, dict(E=1)),
]


@pytest.mark.parametrize("kind,B,Cin,Lin,cfg", CONV_CASES)
def test_conv_family_fwd_bwd(kind, B, Cin, Lin, cfg):
    r = _sandwich(kind, B, Cin, Lin, cfg, seed=B + Cin + Lin)
    m = r["m"]
    Xd, sXd, md = dev(r["X"]), dev(r["sX"]), dev(r["mask"])
    a_in = F.prelu(r["X"], r["sX"])
    pin = _partials_of(a_in).to(DEV)
    rm, rv = torch.zeros(Cin, device=DEV), torch.ones(Cin, device=DEV)
    bn_in = ops.make_bn(pin, 2, B * Lin, rm, rv, update_running=True)
    view = ops.make_view(Xd, sXd, bn_in, md)
    w, bias = dev(m.weight.detach()), dev(m.bias.detach())
    Cout, Lout = r["out"].shape[1], r["out"].shape[2]
    out = torch.empty(B, Cout, Lout, device=DEV)
    pout = torch.zeros(_lib.RAAE_MAX_PARTS, Cout, 2, dtype=torch.float64, device=DEV)
    sOd = dev(r["sO"])
    if kind == "lenlin":
        nout = ops.lenlin_fwd(view, B, Cin, Lin, w, bias, cfg["E"], out, _lib.OUT_STATS_PRELU, sOd, pout)
    else:
        cv = ops.make_conv(Cin, Lin, Cout, Lout, cfg["K"], cfg.get("s", cfg["K"]), cfg.get("p", 0),
                           cfg.get("rep", False), cfg["g"], kind == "convT")
        nout = ops.conv_fwd(view, B, cv, w, bias, out, _lib.OUT_STATS_PRELU, sOd, pout)
    close(out, r["out"], 2e-5, 2e-5, "out")
    a_out = F.prelu(r["out"].detach(), r["sO"])
    tot = pout[:nout].sum(0).cpu()
    close(tot[:, 0], a_out.double().sum((0, 2)), 1e-5, 1e-4, "sum partials")
    close(tot[:, 1], (a_out.double() ** 2).sum((0, 2)), 1e-5, 1e-4, "sumsq partials")
    close(rm, 0.1 * a_in.mean((0, 2)), 1e-4, 1e-6, "running mean")

    # backward: grad spec = (G through BN(PReLU(out)))
    bn_in.update_running = 0
    y = r["y"].detach()
    gp = torch.zeros(2, Cout, 2, dtype=torch.float64)
    h = max(B // 2, 1)
    for i, sl in enumerate((slice(0, h), slice(h, B))):
        gp[i, :, 0] = r["G"][sl].double().sum((0, 2))
        gp[i, :, 1] = (r["G"][sl].double() * y[sl].double()).sum((0, 2))
    bn_out = ops.make_bn(pout, nout, B * Lout)
    go = ops.make_grad(dev(r["G"]), raw=out, slope=sOd, bn=bn_out, g_partials=gp.to(DEV), g_nparts=2)
    nwp, stride = (w.numel() + 63) // 64 * 64, (w.numel() + 63) // 64 * 64 + 128
    slabs = torch.zeros(_lib.RAAE_MAX_PARTS, stride, device=DEV)
    dwv, dbv, dsv = slabs[0, 0:], slabs[0, nwp:], slabs[0, nwp + 64:]
    din = torch.full((B, Cin, Lin), 0.5, device=DEV)
    pdin = torch.zeros(_lib.RAAE_MAX_PARTS, Cin, 2, dtype=torch.float64, device=DEV)
    if kind == "lenlin":
        nsl = ops.lenlin_bwd_weight(go, B, Cin, cfg["E"], view, Lin, dwv, dbv, dsv, stride)
        nd = ops.lenlin_bwd_data(go, B, Cin, cfg["E"], w, view, Lin, din, True, pdin)
    else:
        nsl = ops.conv_bwd_weight(go, B, cv, view, dwv, dbv, dsv, stride)
        nd = ops.conv_bwd_data(go, B, cv, w, view, din, True, pdin)
    tot_s = slabs[:nsl].sum(0)
    dw, db, ds = tot_s[:w.numel()].view(w.shape), tot_s[nwp:nwp + bias.numel()], tot_s[nwp + 64:nwp + 64 + Cout]
    scale = float(r["G"].abs().mean()) * (B * Lout) ** 0.5
    close(dw, m.weight.grad, 5e-4, 5e-5 * scale, "dw")
    close(db, m.bias.grad, 5e-4, 5e-5 * scale, "dbias")
    if kind != "lenlin" or True:
        close(ds, r["dsO"], 5e-4, 5e-5 * scale, "dslope_out")
    close(din, r["dv"] + 0.5, 5e-4, 5e-5, "din (accumulated onto 0.5)")
    tot = pdin[:nd].sum(0).cpu()
    want_din = (r["dv"] + 0.5).double()
    # sums of B*Lin fp32 values each good to ~5e-5: the absolute error of the total grows like sqrt(B*Lin)
    atol = max(1e-3, 5e-5 * (B * Lin) ** 0.5)
    close(tot[:, 0], want_din.sum((0, 2)), 1e-4, atol, "din partial sum")
    close(tot[:, 1], (want_din * r["v"].detach().double()).sum((0, 2)), 1e-4, atol, "din*y partial sum")


@pytest.mark.parametrize("B,C,L,act", [(256, 4, 256, 3), (37, 4, 256, 3), (4099, 4, 256, 4), (130, 8, 64, 0)])
def test_decoder_head_kernels(B, C, L, act):
    """BatchNorm1d(C, affine=False) -> Conv1d(C, 1, 1) -> Softplus(beta=2) / ReLU / nothing (reference model.py:461)
    through the streaming head kernels: forward inside ``raae_conv_fwd``, backward ``raae_head_bwd`` (data gradient,
    its BatchNorm-backward sums and the parameter-gradient slabs in one pass) against torch autograd, and against the
    per-layer kernels they replace (``raae_conv_bwd_data`` + ``raae_conv_bwd_weight``)."""
    torch.manual_seed(B + C)
    X = torch.randn(B, C, L) * 1.7 + 0.3
    conv = torch.nn.Conv1d(C, 1, 1)
    G = torch.randn(B, 1, L)
    # the expectation in float64 (torch's fp32 weight gradient is itself ~1e-2 off on a sum of a million terms)
    conv64 = torch.nn.Conv1d(C, 1, 1).double()
    conv64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    X64 = X.double()
    mean, var = X64.mean((0, 2)), X64.var((0, 2), unbiased=False)
    yhat = ((X64 - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5)).detach().requires_grad_(True)
    pre = conv64(yhat)
    out = F.softplus(pre, beta=2) if act == 3 else (torch.relu(pre) if act == 4 else pre)
    (out * G.double()).sum().backward()
    conv.weight.grad, conv.bias.grad = conv64.weight.grad.float(), conv64.bias.grad.float()

    Xd = dev(X)
    pin = _partials_of(X).to(DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    view = ops.make_view(Xd, None, ops.make_bn(pin, 2, B * L, rm, rv, update_running=True))
    cv = ops.make_conv(C, L, 1, L, 1, 1, 0, False, 1, False)
    w, bias = dev(conv.weight.detach()), dev(conv.bias.detach())
    got = torch.empty(B, 1, L, device=DEV)
    ops.conv_fwd(view, B, cv, w, bias, got, _lib.OUT_RAW, None, None, act)
    close(got, out.detach(), 2e-5, 2e-5, "head forward")
    close(rm, 0.1 * mean.float(), 1e-4, 1e-6, "running mean")

    view = ops.make_view(Xd, None, ops.make_bn(pin, 2, B * L))
    go = ops.make_grad(dev(G), raw=got, act=act)
    assert ops.head_bwd_supported(go, B, cv, view)
    stride = 128
    slabs = torch.zeros(_lib.RAAE_MAX_PARTS, stride, device=DEV)
    din = torch.empty(B, C, L, device=DEV)
    pdin = torch.zeros(_lib.RAAE_MAX_PARTS, C, 2, dtype=torch.float64, device=DEV)
    nd, ns = ops.head_bwd(go, B, cv, w, view, din, pdin, slabs[0, 0:], slabs[0, 64:], stride)
    assert 1 <= nd <= 256 and ns == nd
    tot = slabs[:ns].sum(0)
    scale = float(G.abs().mean()) * (B * L) ** 0.5
    close(tot[:C].view(1, C, 1), conv.weight.grad, 1e-4, 2e-6 * scale, "dw")
    close(tot[64:65], conv.bias.grad, 1e-4, 2e-6 * scale, "dbias")
    close(din, yhat.grad, 1e-5, 1e-6, "din")
    want = yhat.grad.double()
    atol = max(1e-3, 5e-5 * (B * L) ** 0.5)
    ptot = pdin[:nd].sum(0).cpu()
    close(ptot[:, 0], want.sum((0, 2)), 1e-4, atol, "din partial sum")
    close(ptot[:, 1], (want * yhat.detach().double()).sum((0, 2)), 1e-4, atol, "din*y partial sum")

    # the per-layer kernels on the same inputs
    slabs2 = torch.zeros(_lib.RAAE_MAX_PARTS, stride, device=DEV)
    din2 = torch.empty(B, C, L, device=DEV)
    pdin2 = torch.zeros(_lib.RAAE_MAX_PARTS, C, 2, dtype=torch.float64, device=DEV)
    ns2 = ops.conv_bwd_weight(go, B, cv, view, slabs2[0, 0:], slabs2[0, 64:], None, stride)
    nd2 = ops.conv_bwd_data(go, B, cv, w, view, din2, False, pdin2)
    close(din, din2, 1e-6, 1e-7, "din vs per-layer kernel")
    close(tot[:C], slabs2[:ns2].sum(0)[:C], 1e-4, 1e-5 * scale, "dw vs per-layer kernel")
    close(ptot, pdin2[:nd2].sum(0).cpu(), 1e-5, atol, "partials vs per-layer kernel")


@pytest.mark.parametrize("B,Cin,K,s,rep,act", [(1100, 4, 11, 1, True, 0), (2051, 1, 11, 2, True, 0),
                                                (4100, 4, 5, 2, False, 3)])       # 3 = RAAE_OUT_SOFTPLUS
def test_conv_fwd_large_batch_plain_view(B, Cin, K, s, rep, act):
    """Register-blocked large-batch forward without mask / BatchNorm / PReLU on the input and with the
    statistics switched off or on the raw output: the remaining template and flag combinations."""
    g = torch.Generator().manual_seed(B)
    Lin = 256 if K == 11 else 64
    X = torch.randn(B, Cin, Lin, generator=g)
    m = torch.nn.Conv1d(Cin, 4, K, stride=s, padding=(K - 1) // 2, padding_mode="replicate" if rep else "zeros")
    with torch.no_grad():
        ref = m(X)
        if act == 3:
            want = F.softplus(ref, beta=2)
        else:
            want = ref
    Lout = ref.shape[2]
    cv = ops.make_conv(Cin, Lin, 4, Lout, K, s, (K - 1) // 2, rep, 1, False)
    out = torch.empty(B, 4, Lout, device=DEV)
    pout = torch.zeros(_lib.RAAE_MAX_PARTS, 4, 2, dtype=torch.float64, device=DEV)
    kind = _lib.OUT_RAW if act else _lib.OUT_STATS_RAW
    n = ops.conv_fwd(ops.make_view(dev(X)), B, cv, dev(m.weight.detach()), dev(m.bias.detach()), out, kind, None,
                     pout if kind else None, act)
    close(out, want, 2e-5, 2e-5, "out")
    if kind:
        tot = pout[:n].sum(0).cpu()
        close(tot[:, 0], ref.double().sum((0, 2)), 1e-5, 1e-3, "sum partials")
        close(tot[:, 1], (ref.double() ** 2).sum((0, 2)), 1e-5, 1e-3, "sumsq partials")


def test_sum3_and_grad_materialize():
    g = torch.Generator().manual_seed(9)
    B, Cc, L = 24, 4, 64
    A, Bt, Ct = (torch.randn(B, Cc, L, generator=g) for _ in range(3))
    sa, sc = torch.rand(Cc, generator=g) * 0.3, torch.rand(Cc, generator=g) * 0.3
    Ar, Btr, Ctr = (t.clone().requires_grad_(True) for t in (A, Bt, Ct))
    sar = sa.clone().requires_grad_(True)
    yb = F.batch_norm(Btr, None, None, training=True, eps=1e-5)      # identity shortcut: res = bn1(X)
    Y = F.prelu(Ar, sar) + yb + F.prelu(Ctr, sc)
    Y.retain_grad()
    Gn = torch.randn(B, Cc, L, generator=g)
    Rn = F.batch_norm(Y, None, None, training=True, eps=1e-5)
    (Rn * Gn).sum().backward()
    pB = _partials_of(Bt).to(DEV)
    va = ops.make_view(dev(A), dev(sa))
    vb = ops.make_view(dev(Bt), None, ops.make_bn(pB, 2, B * L))
    vc = ops.make_view(dev(Ct), dev(sc))
    y = torch.empty(B, Cc, L, device=DEV)
    pY = torch.zeros(_lib.RAAE_MAX_PARTS, Cc, 2, dtype=torch.float64, device=DEV)
    nY = ops.sum3_fwd(va, vb, vc, B, Cc, L, y, pY)
    close(y, Y, 1e-5, 1e-5, "Y")
    # gradient of term A and of the identity term through the next block's BatchNorm
    gp = torch.zeros(1, Cc, 2, dtype=torch.float64)
    gp[0, :, 0] = Gn.double().sum((0, 2))
    gp[0, :, 1] = (Gn.double() * Rn.detach().double()).sum((0, 2))
    bnY = ops.make_bn(pY, nY, B * L)
    goA = ops.make_grad(dev(Gn), raw=dev(A), slope=dev(sa), bn=bnY, g_partials=gp.to(DEV), g_nparts=1, u=y)
    dA, dsl = torch.empty(B, Cc, L, device=DEV), torch.zeros(64, 64, device=DEV)
    nsl = ops.grad_materialize(goA, B, Cc, L, dA, False, dsl, 64)
    close(dA, Ar.grad, 2e-4, 2e-5, "dA")
    close(dsl[:nsl, :Cc].sum(0), sar.grad, 5e-4, 1e-4, "dslope A")
    goI = ops.make_grad(dev(Gn), bn=bnY, g_partials=gp.to(DEV), g_nparts=1, u=y)
    acc = torch.ones(B, Cc, L, device=DEV)
    ops.grad_materialize(goI, B, Cc, L, acc, True, None, 0)
    close(acc, Y.grad + 1.0, 2e-4, 2e-5, "identity shortcut gradient")


@pytest.mark.parametrize("n,k", [(1050, 6), (37, 3), (3, 2), (4100, 13), (20000, 2)])
def test_style_metrics_match_scipy(n, k):
    """raae_style_metrics == scipy.stats.shapiro / spearmanr (what the reference calls on the host copy of the
    validation styles, trainer.py:286-292), including tied values, to 1e-12 (f64 sums in another order)."""
    import itertools
    from scipy.stats import shapiro, spearmanr
    from rankaae_amd.metrics import StyleMetrics
    rng = np.random.default_rng(n + k)
    z = rng.standard_normal((n, k)).astype(np.float32)
    z[:, 0] += 0.5 * z[:, k - 1]                              # a correlated pair
    if n > 3:
        z[:, 1] = np.round(z[:, 1] * 4) / 4                   # heavy ties
        z[::7, k - 1] = z[0, k - 1]                           # a run of equal values
    m = StyleMetrics(n, k, DEV)
    m.launch(dev(torch.from_numpy(z)))
    W, rho = m.read()
    W_ref = np.array([shapiro(z[:, c]).statistic for c in range(k)])
    rho_ref = np.array([spearmanr(z[:, p], z[:, q]).correlation for p, q in itertools.combinations(range(k), 2)])
    assert W.shape == W_ref.shape and rho.shape == rho_ref.shape
    assert np.abs(W - W_ref).max() < 1e-12, (W, W_ref)
    assert np.abs(rho - rho_ref).max() < 1e-12, (rho, rho_ref)
    m.launch(dev(torch.from_numpy(z)))                        # fixed-order sums: bitwise repeatable
    W2, rho2 = m.read()
    assert np.array_equal(W, W2) and np.array_equal(rho, rho2)


@pytest.mark.parametrize("n_real,n_fake,ns,drop", [(40, 23, 6, True), (256, 256, 6, True), (64, 36, 13, False),
                                                   (4096, 4100, 6, True)])
def test_disc_fused_matches_autograd(n_real, n_fake, ns, drop):
    """raae_disc_fused (the adversarial branch in one launch) against torch autograd of the same computation:
    [z_real; styles] + sigma*noise -> Linear/PReLU/Dropout x2 -> Linear -> BCE-with-logits(ones | zeros); loss,
    every parameter gradient (fixed-order slab sum) and dstyles = -alpha * dL/dstyles."""
    g = torch.Generator().manual_seed(n_real + ns)
    n, H, sigma, alpha, p = n_real + n_fake, 64, 0.56, 0.37, 0.056
    lin = [torch.nn.Linear(ns, H), torch.nn.Linear(H, H), torch.nn.Linear(H, 1)]
    pre = [torch.nn.PReLU(H), torch.nn.PReLU(H)]
    with torch.no_grad():
        for q in pre:
            q.weight.copy_(0.1 + 0.3 * torch.rand(H, generator=g))
    z_real = torch.randn(n_real, ns, generator=g)
    styles = torch.randn(n_fake, ns, generator=g).requires_grad_(True)
    noise = torch.randn(n, ns, generator=g)
    masks = [((torch.rand(n, H, generator=g) > p).float() / (1 - p)) if drop else None for _ in range(2)]
    x = torch.cat([z_real, styles]) + sigma * noise
    h = x
    for i in range(2):
        h = pre[i](lin[i](h))
        if masks[i] is not None:
            h = h * masks[i]
    o = lin[2](h).squeeze(1)
    loss = F.binary_cross_entropy_with_logits(o[:n_real], torch.ones(n_real)) + \
        F.binary_cross_entropy_with_logits(o[n_real:], torch.zeros(n_fake))
    loss.backward()

    class L:
        pass
    layers = []
    for i in range(3):
        l = L()
        l.w, l.b, l.N = dev(lin[i].weight.detach()), dev(lin[i].bias.detach()), lin[i].out_features
        l.prelu = L()
        l.prelu.weight = dev(pre[i].weight.detach()) if i < 2 else None
        layers.append(l)
    params = [layers[0].w, layers[0].b, layers[0].prelu.weight, layers[1].w, layers[1].b, layers[1].prelu.weight,
              layers[2].w, layers[2].b]
    offs, tot = {}, 0
    for q in params:
        offs[id(q)] = tot
        tot += (q.numel() + 63) // 64 * 64
    slabs = torch.zeros(256, tot, device=DEV)
    dstyles = torch.empty(n_fake, ns, device=DEV)
    out = torch.zeros(1, device=DEV)
    nsl = ops.disc_fused(dev(z_real), dev(styles.detach()), dev(noise), sigma, dev(masks[0]) if drop else None,
                         dev(masks[1]) if drop else None, layers, torch.tensor([alpha], device=DEV), n_real, n_fake, ns,
                         lambda q: slabs[0, offs[id(q)]:], tot, dstyles, torch.zeros(256, dtype=torch.float64, device=DEV),
                         torch.zeros(1, dtype=torch.int32, device=DEV), out)
    assert 1 <= nsl <= 256
    close(out, loss.detach().view(1), 1e-5, 1e-6, "loss")
    # A hidden unit whose pre-activation is pure rounding residue can take the other PReLU slope here than in the torch
    # reference (another summation order).  That is VERIFIED, not assumed: the rows that hold such a unit are found
    # from a float64 repeat of the forward (|z| < 1e-5 max|z| in either hidden layer); only those rows may miss the
    # tight tolerance, and without any such row nothing may.
    with torch.no_grad():
        z1 = x.detach().double() @ lin[0].weight.double().T + lin[0].bias.double()
        a1 = torch.where(z1 > 0, z1, z1 * pre[0].weight.double())
        if masks[0] is not None:
            a1 = a1 * masks[0].double()
        z2 = a1 @ lin[1].weight.double().T + lin[1].bias.double()
        near = ((z1.abs() < 1e-5 * z1.abs().max()).any(1) | (z2.abs() < 1e-5 * z2.abs().max()).any(1)).numpy()
    near_rows = set(int(i) - n_real for i in np.nonzero(near)[0] if i >= n_real)
    n_near = int(near.sum())
    want = (-alpha * styles.grad).double().numpy()
    err = np.abs(dstyles.cpu().double().numpy() - want)
    tol = 1e-7 + 2e-4 * float(np.abs(want).max()) + 2e-4 * np.abs(want)
    bad_rows = set(int(i) for i in np.unique(np.nonzero(err > tol)[0]))
    assert bad_rows <= near_rows, (sorted(bad_rows), sorted(near_rows), err.max())
    assert all((err[r_] <= 0.5 * np.abs(want).max()).all() for r_ in bad_rows)
    ref = [lin[0].weight.grad, lin[0].bias.grad, pre[0].weight.grad, lin[1].weight.grad, lin[1].bias.grad,
           pre[1].weight.grad, lin[2].weight.grad, lin[2].bias.grad]
    for q, r, name in zip(params, ref, ["dw1", "db1", "ds1", "dw2", "db2", "ds2", "dw3", "db3"]):
        got = slabs[:nsl, offs[id(q)]:offs[id(q)] + q.numel()].sum(0).view(r.shape)
        # (a flipped unit of row r reaches, through that row, one row of dW2 and every entry of dW1: up to ~130 entries
        # per flip may miss the tight tolerance, by a few percent of the tensor's largest entry at most)
        e = (got.cpu().double() - r.double()).abs().numpy()
        rmax = float(r.abs().max())
        t = 1e-7 + 2e-4 * rmax + 2e-4 * r.abs().double().numpy()
        n_bad = int((e > t).sum())
        assert n_bad <= n_near * 130 and (n_bad == 0 or e.max() <= 0.05 * rmax + 1e-7), (name, n_bad, n_near, e.max(), rmax)


@pytest.mark.parametrize("n_real,n_fake", [(1024, 1024), (4096, 4100)])
def test_disc_fused_matrix_core_form(n_real, n_fake):
    """From 2048 rows (real + fake) the three 64 x 64 contractions of raae_disc_fused run on the matrix cores
    (v_mfma_f32_16x16x4_f32, the MM instance of the kernel); below that the VALU instance runs, which the cases above
    pin at 63 .. 512 rows.  The matrix-core instance at 2048 and 8196 rows against torch autograd, same tolerances."""
    test_disc_fused_matches_autograd(n_real, n_fake, 6, True)


def test_loss_kernels_finish_in_kernel():
    """``fin=...``: the last workgroup to arrive sums the partials inside the loss kernel; the value (and the
    accumulating slot) equal what the separate raae_loss_finalize launch gives, bit for bit, call after call."""
    from oracle.ref_model import gaussian_taps
    g = torch.Generator().manual_seed(3)
    B, L = 256, 256
    x, y = dev(torch.rand(B, L, generator=g) + 0.2), dev(torch.rand(B, L, generator=g) + 0.2)
    part = torch.zeros(_lib.RAAE_MAX_PARTS, dtype=torch.float64, device=DEV)
    ticket = torch.zeros(1, dtype=torch.int32, device=DEV)
    ref, got = torch.zeros(8, device=DEV), torch.zeros(8, device=DEV)
    taps = gaussian_taps(17, 3.0).tolist()
    for rep in range(3):
        n = ops.recon_loss_fwd_bwd(x, y, B, L, True, part, None)
        ops.loss_finalize(part, n, 1.0, ref, 2)
        ops.recon_loss_fwd_bwd(x, y, B, L, True, part, None, fin=(1.0, got, 2, -1, ticket))
        n = ops.smooth_loss_fwd_bwd(y, B, L, taps, part, None)
        ops.loss_finalize(part, n, 1.0, ref, 4)
        ops.smooth_loss_fwd_bwd(y, B, L, taps, part, None, fin=(1.0, got, 4, -1, ticket))
        n = ops.mse_fwd_bwd(x, y, B * L, part, None)
        ops.loss_finalize(part, n, 1.0, ref, 3, 5)
        ops.mse_fwd_bwd(x, y, B * L, part, None, fin=(1.0, got, 3, 5, ticket))
        assert torch.equal(ref, got), (rep, ref, got)
        assert int(ticket) == 0
    assert float(got[5]) > 2.5 * float(got[3]) > 0           # three accumulations
