"""CPU oracle: the loss functions and the five-phase training step of the reference.

TEST INFRASTRUCTURE (see ``oracle/ref_model.py`` header for who may import it).
Plain PyTorch-fp32 autograd on the CPU; every function cites what it restates.
All random draws go through the global torch CPU generator in the reference's order
(SURVEY.md §3.4), so with the same seed this reproduces the reference bit for bit on
the same machine/thread count -- that is what ``tests/test_oracle_golden.py`` pins.
"""
import itertools

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from .ref_model import AE_CLASSES, DiscriminatorFC, gaussian_taps


# ----------------------------------------------------------------------------- losses
def kendall_constraint(descriptors, styles, activate=False):
    """Pairwise sign-concordance loss, ``sc/utils/functions.py:37-79``.

    Restated literally (materialised [B,B,n_aux]; boolean-mask in-place scaling) so
    that the arithmetic -- including its float32 summation -- is the reference's.
    """
    n_aux = styles.shape[1]
    target = torch.sign(descriptors[:, None, :] - descriptors[None, :, :])
    pred = styles[:, None, :] - styles[None, :, :]
    n = pred.size(0)
    product = pred * target
    if activate:
        same = product > 0
        opp = product < 0
        idx = torch.arange(n_aux)
        for k in range(n_aux):
            sel = idx == k
            n_same = max(torch.numel(product[same & sel]), 1)
            n_opp = max(torch.numel(product[opp & sel]), 1)
            product[same & sel] *= n_opp / max(n_same, n_opp)
    return -product.sum() / ((n ** 2 - n) * n_aux)


def kendall_closed_form(descriptors, styles, activate=False):
    """One-pass closed form of the same loss and its gradient,
    float64.  Returns ``(loss, dstyles)``.  This is the algorithm the HIP kernel
    implements; ``tests/test_oracle_golden.py`` checks it against the literal form."""
    d = descriptors.double()
    z = styles.double()
    n, k = z.shape
    s = torch.sign(d[:, None, :] - d[None, :, :])
    p = (z[:, None, :] - z[None, :, :]) * s
    pos, neg = p > 0, p < 0
    n_same = pos.sum((0, 1)).clamp(min=1).double()
    n_opp = neg.sum((0, 1)).clamp(min=1).double()
    c = n_opp / torch.maximum(n_same, n_opp) if activate else torch.ones(k, dtype=torch.float64)
    norm = (n * n - n) * k
    s_pos = (p * pos).sum((0, 1))
    s_neg = (p * neg).sum((0, 1))
    loss = -(c * s_pos + s_neg).sum() / norm
    g_pos = (s * pos).sum(1)
    g_neg = (s * neg).sum(1)
    return loss, -(2.0 / norm) * (c * g_pos + g_neg)


def recon_loss(spec_in, spec_out, scale=False):
    """``sc/utils/functions.py:81-107``."""
    spec_in = spec_in.clone()
    if not scale:
        return F.mse_loss(spec_out, spec_in)
    ratio = torch.abs(spec_out.mean(dim=1)) / torch.abs(spec_in.mean(dim=1))
    loss = ((ratio - 1.0) ** 2).mean() * 0.1
    ratio = torch.clamp(ratio.detach(), min=0.7, max=1.3)
    return loss + F.mse_loss(spec_out, (spec_in.T * ratio).T)


def adversarial_loss(spec_in, styles, disc, alpha, batch_size):
    """``sc/utils/functions.py:109-132`` (BCE-with-logits; real = N(0,I) of the
    *configured* batch size, fake = encoder styles of the actual batch)."""
    nstyle = styles.size(1)
    z_real = torch.randn(batch_size, nstyle, requires_grad=True)
    real_pred = disc(z_real, alpha)
    fake_pred = disc(styles, alpha)
    ones = torch.ones(batch_size, dtype=torch.float32)
    zeros = torch.zeros(spec_in.size(0), dtype=torch.float32)
    bce = nn.BCEWithLogitsLoss()
    return bce(real_pred.squeeze(), ones) + bce(fake_pred.squeeze(), zeros)


def mutual_info_loss(spec_in, styles, encoder, decoder):
    """``sc/utils/functions.py:174-192``."""
    z = torch.randn(spec_in.size(0), styles.size(1), requires_grad=False)
    return F.mse_loss(encoder(decoder(z)), z)


def smoothness_loss(spec_out, gs_kernel_size=17):
    """``sc/utils/functions.py:194-212`` + ``GaussianSmoothing`` (``model.py:177-229``):
    replicate-pad, 17-tap normalised Gaussian (sigma 3), MSE(x, smoothed x)."""
    w = gaussian_taps(gs_kernel_size, 3.0).view(1, 1, -1).to(spec_out.dtype)   # fp32 taps (a no-op for fp32 input)
    pad = (gs_kernel_size - 1) // 2
    padded = F.pad(spec_out.unsqueeze(1), (pad, pad), mode="replicate")
    smoothed = F.conv1d(padded, w, groups=1).squeeze(1)
    return F.mse_loss(spec_out, smoothed)


def alpha(epoch_percentage, step=800, limit=0.7):
    """``sc/utils/functions.py:214-219`` (float64 numpy)."""
    return (2. / (1. + np.exp(-1.0E4 / step * epoch_percentage)) - 1) * limit


# ----------------------------------------------------------------------------- data
def split_rows(n_rows, ratios=(0.7, 0.15, 0.15)):
    """Contiguous train/val/test split of ``sc/clustering/dataloader.py:14-20``."""
    n = [int(n_rows * r) for r in ratios]
    n[-1] = n_rows - sum(n[:-1])
    return n


def epoch_permutation(n_train):
    """Row order of one epoch of ``DataLoader(shuffle=True, num_workers=0)``
    (SURVEY.md finding 9): the iterator draws ``_base_seed`` (one int64 ``random_()``
    from the global generator), the RandomSampler draws its own seed (a second one) and
    permutes with a private generator."""
    torch.empty((), dtype=torch.int64).random_()  # _base_seed
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n_train, generator=g)


# ----------------------------------------------------------------------------- trainer
class OracleTrainer:
    """Restates ``Trainer.from_data`` + ``Trainer.train`` (``sc/clustering/trainer.py:38-474``)
    for the reachable configuration set (SURVEY.md finding 4): ``ae_form`` in {FC, compact},
    gradient reversal, ``DiscriminatorFC``, Adam/AdamW, ``n_aux >= 1``.

    ``spec``/``aux`` are the full float64 arrays (what the reference reads from CSV).
    """

    metric_weights = [1.0, -1.0, -0.01, -1.0, -1.0]
    gau_kernel_size = 17

    def __init__(self, spec, aux, cfg, anomaly=False):
        self.cfg = dict(cfg)
        c = self.cfg
        n_train, n_val, _ = split_rows(len(spec))
        self.train_spec = np.asarray(spec[:n_train])
        self.train_aux = np.asarray(aux[:n_train])
        self.val_spec = torch.tensor(np.asarray(spec[n_train:n_train + n_val]), dtype=torch.float32)
        self.val_aux = torch.tensor(np.asarray(aux[n_train:n_train + n_val]), dtype=torch.float32)
        enc_cls, dec_cls = AE_CLASSES[c["ae_form"]]
        # construction order enc -> dec -> D fixes the init draws (trainer.py:442-463)
        self.encoder = enc_cls(nstyle=c["nstyle"], dropout_rate=c["dropout_rate"], dim_in=c["dim_in"],
                               n_layers=c["n_layers"])
        self.decoder = dec_cls(nstyle=c["nstyle"], dropout_rate=c["dropout_rate"],
                               last_layer_activation=c["decoder_activation"], dim_out=c["dim_out"],
                               n_layers=c["n_layers"])
        self.discriminator = DiscriminatorFC(nstyle=c["nstyle"], dropout_rate=c["dis_dropout_rate"],
                                             noise=c["dis_noise"], layers=c["FC_discriminator_layers"])
        self.epoch_stop_smooth = c.get("epoch_stop_smooth", 500)
        self._load_optimizers()
        self.anomaly = anomaly
        self.last = {}
        self.phase_hook = None   # callable(phase_name) invoked after backward, before optimizer.step
        self.post_hook = None    # callable(phase_name) invoked right after optimizer.step

    # trainer.py:333-408
    def _load_optimizers(self):
        c = self.cfg
        cls = {"Adam": torch.optim.Adam, "AdamW": torch.optim.AdamW}[c["optimizer_name"]]
        enc, dec, dis = self.encoder, self.decoder, self.discriminator
        lr = c["lr_base"]
        betas_d = (c["dis_beta"] * 0.9, c["dis_beta"] * 0.009 + 0.99)
        self.optimizers = {
            "reconstruction": cls([{"params": enc.parameters()}, {"params": dec.parameters()}],
                                  lr=c["lr_ratio_Reconn"] * lr, weight_decay=c["weight_decay"]),
            "mutual_info": cls([{"params": enc.parameters()}, {"params": dec.parameters()}],
                               lr=c["lr_ratio_Mutual"] * lr),
            "smoothness": cls([{"params": dec.parameters()}], lr=c["lr_ratio_Smooth"] * lr,
                              weight_decay=c["weight_decay"]),
            "correlation": cls([{"params": enc.parameters()}], lr=c["lr_ratio_Corr"] * lr,
                               weight_decay=c["weight_decay"]),
            "adversarial": cls([{"params": dis.parameters()}, {"params": enc.parameters()}],
                               lr=c["lr_ratio_dis"] * lr, betas=betas_d),
        }
        self.schedulers = {k: torch.optim.lr_scheduler.ReduceLROnPlateau(
            o, mode="min", factor=c["sch_factor"], patience=c["sch_patience"], cooldown=0, threshold=0.01)
            for k, o in self.optimizers.items()}

    def zerograd(self):
        for m in (self.encoder, self.decoder, self.discriminator):
            m.zero_grad()

    def _finish(self, name):
        if self.phase_hook is not None:
            self.phase_hook(name)
        self.optimizers[name].step()
        if self.post_hook is not None:
            self.post_hook(name)

    def step_phases(self, spec_in, aux_in, alpha_, epoch=0, kendall=None):
        """The five phases of one batch (trainer.py:103-204) as a generator: after each phase's ``backward`` it
        yields ``(optimizer name, loss)`` and expects the caller to apply that optimizer before resuming
        (``train_step`` does exactly that; ``train_step_sharded`` first averages the gradients of several replicas).
        ``kendall`` not None: before the rank loss the generator yields ``("correlation_styles", (aux, styles))`` and
        expects the loss tensor back through ``send`` (data-parallel runs with global pairs)."""
        c = self.cfg
        enc, dec, dis = self.encoder, self.decoder, self.discriminator
        n_aux = aux_in.size(-1)
        spec_in += torch.randn_like(spec_in, requires_grad=False) * c["spec_noise"]
        styles = enc(spec_in)
        dec(styles)  # result unused by the reference too, but advances BN stats and the RNG
        self.zerograd()
        loss = adversarial_loss(spec_in, styles, dis, alpha_, batch_size=c["batch_size"])
        loss.backward()
        yield "adversarial", loss
        self.zerograd()
        styles = enc(spec_in)
        if kendall is None:
            loss = kendall_constraint(aux_in, styles[:, :n_aux], activate=c["kendall_activation"])
        else:       # the caller forms the rank loss (it needs the styles of the other replicas) and sends it back
            loss = yield "correlation_styles", (aux_in, styles[:, :n_aux])
        loss.backward()
        yield "correlation", loss
        self.zerograd()
        loss = recon_loss(spec_in, dec(enc(spec_in)), scale=c["use_flex_spec_target"])
        loss.backward()
        yield "reconstruction", loss
        self.zerograd()
        styles = enc(spec_in)
        loss = mutual_info_loss(spec_in, styles, enc, dec)
        loss.backward()
        yield "mutual_info", loss
        if epoch < self.epoch_stop_smooth:
            self.zerograd()
            loss = smoothness_loss(dec(enc(spec_in)), self.gau_kernel_size)
            loss.backward()
            yield "smoothness", loss
        self.zerograd()

    LOSS_KEY = {"adversarial": "adversarial", "correlation": "kendall", "reconstruction": "recon",
                "mutual_info": "mutual_info", "smoothness": "smooth"}

    # trainer.py:103-204 -- one batch through the five phases
    def train_step(self, spec_in, aux_in, alpha_, epoch=0):
        out = {"smooth": torch.tensor(0)}
        for name, loss in self.step_phases(spec_in, aux_in, alpha_, epoch):
            self._finish(name)
            out[self.LOSS_KEY[name]] = loss
        self.last = {k: float(v.detach()) for k, v in out.items()}
        return self.last

    # trainer.py:206-268
    def validate(self, alpha_):
        c = self.cfg
        enc, dec, dis = self.encoder, self.decoder, self.discriminator
        for m in (enc, dec, dis):
            m.eval()
        torch.empty((), dtype=torch.int64).random_()  # val DataLoader iterator's _base_seed draw
        z = enc(self.val_spec)
        spec_out = dec(z)
        n_aux = self.val_aux.size(-1)
        val = {
            "recon": recon_loss(self.val_spec, spec_out),
            "kendall": kendall_constraint(self.val_aux, z[:, :n_aux], activate=c["kendall_activation"]),
            "smooth": smoothness_loss(spec_out, self.gau_kernel_size),
            "mutual_info": mutual_info_loss(self.val_spec, z, enc, dec),
            "adversarial": adversarial_loss(self.val_spec, z, dis, alpha_, batch_size=c["batch_size"]),
        }
        return z, {k: float(v.detach()) for k, v in val.items()}

    def train(self, max_epoch=None, callback=None, record=None):
        """Full loop incl. validation + scipy metrics + schedulers; ``record`` (dict of
        lists) receives every loss call in the order the golden fixtures store them."""
        from scipy.stats import shapiro, spearmanr
        c = self.cfg
        max_epoch = c["max_epoch"] if max_epoch is None else max_epoch
        metrics = None
        prev_anomaly = torch.is_anomaly_enabled()
        torch.autograd.set_detect_anomaly(self.anomaly)
        n_train = len(self.train_spec)
        bs = c["batch_size"]
        for epoch in range(max_epoch):
            for m in (self.encoder, self.decoder, self.discriminator):
                m.train()
            alpha_ = alpha(epoch / max_epoch, c["alpha_flat_step"], c["alpha_limit"])
            perm = epoch_permutation(n_train).numpy()
            n_batch = -(-n_train // bs)
            avg_mi = 0.0
            for ib in range(n_batch):
                rows = perm[ib * bs:(ib + 1) * bs]
                spec_in = torch.tensor(self.train_spec[rows], dtype=torch.float32)
                aux_in = torch.tensor(self.train_aux[rows], dtype=torch.float32)
                losses = self.train_step(spec_in, aux_in, alpha_, epoch)
                avg_mi += losses["mutual_info"]
                if record is not None:
                    for k in ("adversarial", "kendall", "recon", "mutual_info"):
                        record[k].append(losses[k])
                    if epoch < self.epoch_stop_smooth:
                        record["smooth"].append(losses["smooth"])
            z, val = self.validate(alpha_)
            if record is not None:
                for k in ("recon", "kendall", "smooth", "mutual_info", "adversarial"):
                    record[k].append(val[k])
            avg_mi /= n_batch
            style_np = z.detach().clone().numpy().T
            sh = [shapiro(x).statistic for x in style_np]
            coupling = np.max(np.fabs([spearmanr(style_np[a], style_np[b]).correlation
                                       for a, b in itertools.combinations(range(style_np.shape[0]), 2)]))
            metrics = [min(sh), val["recon"], avg_mi, coupling, val["kendall"]]
            combined = -(np.array(self.metric_weights) * np.array(metrics)).sum()
            for sch in self.schedulers.values():
                sch.step(combined)
            if callback is not None:
                callback(epoch, metrics)
        torch.autograd.set_detect_anomaly(prev_anomaly)
        return metrics


# ----------------------------------------------------------------------------- data parallel emulation
def train_step_sharded(replicas, shards, alpha_, rng_states, epoch=0, global_pairs=False, hook=None):
    """Synchronous data parallelism over ``W = len(replicas)`` ranks, emulated sequentially ("parity
    under DP"): ``replicas`` are ``OracleTrainer``s holding the SAME weights and optimizer state, ``shards[r] =
    (spec, aux)`` is rank r's slice of the global batch, ``rng_states[r]`` the state of rank r's own generator
    (updated in place).  Per phase every replica runs forward + backward on its shard -- per-replica BatchNorm batch
    statistics and, by default, the rank loss over the pairs INSIDE the shard (DDP semantics) -- the parameter
    gradients are averaged over the replicas, and every replica applies the averaged gradient (so the replicas stay
    identical).  ``global_pairs``: the rank loss runs over all pairs of the global batch instead (what the
    reference's single-process loss on that batch computes, functions.py:63-77); each replica differentiates the
    global loss w.r.t. its own rows only and the contribution is scaled by W, so that the AVERAGE over replicas is
    the gradient of the global loss.  ``hook(name, local_grads[r][i], mean_grads[i])`` sees every phase.
    Returns the per-replica loss dictionaries."""
    W = len(replicas)
    outs = [{"smooth": 0.0} for _ in range(W)]

    gens = []
    for r, tr in enumerate(replicas):
        torch.set_rng_state(rng_states[r])
        gens.append(tr.step_phases(shards[r][0], shards[r][1], alpha_, epoch, kendall=True if global_pairs else None))
        rng_states[r] = torch.get_rng_state()
    n_phases = 5 if epoch < replicas[0].epoch_stop_smooth else 4
    for ph in range(n_phases):
        names, losses = [], []
        pending = []
        for r, g in enumerate(gens):
            torch.set_rng_state(rng_states[r])
            item = next(g)
            rng_states[r] = torch.get_rng_state()
            if item[0] == "correlation_styles":
                pending.append(item[1])
            else:
                names.append(item[0])
                losses.append(item[1])
        if pending:
            # every replica's phase-B styles exist now: rank r differentiates the GLOBAL loss w.r.t. its own rows
            # (the other shards' rows enter as constants), scaled by W so that the average over ranks is the
            # gradient of the global loss
            all_aux = torch.cat([a_ for a_, _ in pending])
            act = replicas[0].cfg["kendall_activation"]
            for r, g in enumerate(gens):
                parts = [z_ if q == r else z_.detach() for q, (_, z_) in enumerate(pending)]
                loss = W * kendall_constraint(all_aux, torch.cat(parts), activate=act)
                name, loss_back = g.send(loss)
                names.append(name)
                losses.append(loss_back / W)            # the loss every rank reports is the global one
        assert len(set(names)) == 1
        name = names[0]
        plists = [[p for grp in tr.optimizers[name].param_groups for p in grp["params"]] for tr in replicas]
        local = [[None if p.grad is None else p.grad.detach().clone() for p in pl] for pl in plists]
        mean = []
        for i in range(len(plists[0])):
            gs = [local[r][i] for r in range(W)]
            mean.append(None if gs[0] is None else torch.stack(gs).sum(0) / W)
        for r in range(W):
            for p, g_ in zip(plists[r], mean):
                if g_ is not None:
                    p.grad = g_.clone()
        if hook is not None:
            hook(name, local, mean)
        for r, tr in enumerate(replicas):
            tr._finish(name)
            outs[r][OracleTrainer.LOSS_KEY[name]] = float(losses[r].detach())
    for r, g in enumerate(gens):
        torch.set_rng_state(rng_states[r])
        for _ in g:
            raise AssertionError("more phases than expected")
        rng_states[r] = torch.get_rng_state()
    return outs


# ----------------------------------------------------------------------------- derived parity bounds
GRAD_NOISE = 1e-5     # |g|_inf-relative distance of two correct fp32 evaluations of one gradient (DESIGN.md section 7:
#                       the reference's own fp32 gradient is 4.7e-5 from float64 arithmetic behind the first BatchNorm)


def first_step(cfg, model_seed, spec, aux, mode=0, alpha_=0.0):
    """The oracle's FIRST training step from ``model_seed`` under a perturbation of rounding-error size:
      mode  0        none
      mode +1 / -1   every input spectrum value one float32 ulp up / down
      mode  2..9     a random up/down one-ulp pattern over the inputs (seeded with ``mode``)
      mode >= 10     every parameter gradient g of every phase becomes g + GRAD_NOISE * |g|_inf * N(0, 1) before the
                     optimizer step (private generator seeded with ``mode``; the global generator is not touched) --
                     what a second fp32 implementation with another summation order hands to Adam."""
    s32 = np.asarray(spec, dtype=np.float32)
    up, down = np.nextafter(s32, np.float32(np.inf)), np.nextafter(s32, np.float32(-np.inf))
    if mode == 1:
        s32 = up
    elif mode == -1:
        s32 = down
    elif 2 <= mode < 10:
        s32 = np.where(np.random.default_rng(mode).integers(0, 2, s32.shape).astype(bool), up, down)
    torch.manual_seed(model_seed)
    tr = OracleTrainer(s32.astype(np.float64), aux, cfg)
    for m in (tr.encoder, tr.decoder, tr.discriminator):
        m.train()
    if mode >= 10:
        gen = torch.Generator().manual_seed(mode)

        def noisy(name):
            for grp in tr.optimizers[name].param_groups:
                for p in grp["params"]:
                    if p.grad is not None:
                        p.grad.add_(torch.randn(p.grad.shape, generator=gen) * (GRAD_NOISE * float(p.grad.abs().max())))
        tr.phase_hook = noisy
    rows = epoch_permutation(len(tr.train_spec)).numpy()[:cfg["batch_size"]]
    epoch = 0 if 0 < cfg.get("epoch_stop_smooth", 500) else 10 ** 9
    return tr.train_step(torch.tensor(tr.train_spec[rows], dtype=torch.float32),
                         torch.tensor(tr.train_aux[rows], dtype=torch.float32), alpha_, epoch)


def derived_bounds(cfg, model_seed, spec, aux, keys, modes=(1, -1, 2, 3, 11, 12, 13, 14), alpha_=0.0):
    """``(oracle losses, {key: bound})`` for a first-step comparison.  The losses of the later phases follow Adam's
    first, sign-like updates (step = lr * g / (|g| + eps): an entry whose gradient is rounding noise moves by +-lr)
    and amplify rounding noise (SURVEY finding 8); how much is MEASURED, not guessed: the bound is 3x the largest
    move of the oracle's own loss over eight perturbations of rounding-error size (``first_step``), plus a 1e-4
    relative floor."""
    base = first_step(cfg, model_seed, spec, aux, 0, alpha_)
    pert = [first_step(cfg, model_seed, spec, aux, m, alpha_) for m in modes]
    return base, {k: 3.0 * max(abs(p[k] - base[k]) for p in pert) + 1e-4 * abs(base[k]) + 1e-7 for k in keys}


CONSTRAINING_REL = 0.05


def constraining(base, bound, keys, rel=CONSTRAINING_REL):
    """``(tight, loose)``: the keys whose derived bound is at most ``rel`` of the oracle's value, and the others.  A
    derived bound of 30-50 % (the mutual-information loss of the dense networks after Adam's first sign-like updates)
    is an honest statement of the step's sensitivity but asserts nothing: such keys are REPORTED as not constraining
    and left to the teacher-forced (P2) and frozen-weight (P4) comparisons, which pin those phases to 1e-4."""
    tight = [k for k in keys if bound[k] <= rel * abs(base[k]) + 1e-7]
    return tight, [k for k in keys if k not in tight]
