"""Training-step parity on the GPU (SURVEY.md 8d protocol).

P1  step 1 against the REFERENCE's own values in ``tests/golden/ref_*.json``: the adversarial
    and rank losses (computed before any ill-conditioned update) rel <= 1e-4.  The three later
    phase losses are evaluated AFTER Adam's first, sign-like updates (step = lr*g/(|g|+eps)), which
    amplify rounding noise: the bound for them is DERIVED, not guessed -- the oracle's first step is
    repeated with the input spectra moved by one float32 ulp (all up, all down, two random patterns)
    and HIP must lie within 3x the largest move of the oracle's own loss (+ 1e-4 relative floor).
    Measured in the build container: recon moves by up to 1e-3, mutual-info 3e-2, smoothness 5e-2
    relative under such a perturbation (dense networks); 1e-6 / 5e-3 / 1e-2 for the conv networks.
P4  FREE-RUNNING over two whole epochs against the reference's ``ref_*_frozen.json`` (lr_base = 0:
    Adam/AdamW then leave every weight in place, so the trajectory is not chaotic while every
    kernel of the five phases, the random tape, the BatchNorm running statistics and the per-epoch
    validation still run): every training loss of every step incl. the ragged last batch, every
    validation loss, the model-selection metrics, the final running statistics and eval-mode styles.
P2  teacher-forced, at PHASE granularity: the oracle's state (weights, BN statistics, Adam
    moments, step counts) is loaded into the HIP engine before step k, and again after every
    phase's optimizer step; both run on the same batch with the same random tape.  The five
    losses must agree to rel <= 1e-4 and every phase gradient to |dg|_inf <= 5e-3 |g|_inf (+ 1e-5 of
    the phase's largest gradient entry, for tensors whose gradient is pure rounding noise) per tensor.  Why not 1e-4 on gradients: both sides are fp32 with different
    summation orders, and BatchNorm over features whose batch variance is tiny (PReLU slope 0.01
    on all-negative pre-activations: std ~7e-5) amplifies 1e-6 rounding noise ~300x; measured on
    the first FC layer: torch-fp32 vs fp64 4.7e-5, HIP vs fp64 5.5e-5 after the BatchNorm.
    The rank loss is additionally DISCONTINUOUS in the styles (its weights c_k are ratios of
    pair COUNTS, functions.py:73-75): one pair changing sign moves the loss by ~1e-2 at B=64.
    When the tight comparison fails we therefore require instead (a) styles within 5e-4 of the
    oracle's and (b) the HIP loss == the oracle's loss function evaluated on the HIP styles
    (rel 1e-5), and compare that phase's gradients at 2e-2.
    At BASELINE configs[2]'s batch (4096 rows; cases ``*_b4096``) the reference's fp32 batch reductions are
    themselves 0.5-1 % away from exact arithmetic on the cancelling gradients (adversarial and smoothness
    phases), and the kernels here sum in another order: a tensor that misses 5e-3 (in any case) is judged against a
    FLOAT64 repeat of the same oracle step (same parameters, inputs and replayed random tensors,
    ``oracle_float64_gradients``): HIP's distance from it must be within 3x the reference-fp32 distance.
    One more legitimate discontinuity: a PReLU input that is rounding noise around zero picks the other slope in
    one of two fp32 implementations; that moves one sample's gradient at one unit by the slope ratio and every weight
    gradient upstream of it by ~1/sqrt(B) of its size.  At BASELINE configs[0]'s batch (``fc_example``: 1024 rows x 64
    units x 8 layers, pre-activations 1e-6 ... 4e-5 of their scale apart between the two forwards) a phase holds a
    handful of such entries.  They are not excused but ACCOUNTED for (``kink_adjusted``): the entries where HIP's
    stored pre-activation and the oracle's lie on different sides of zero, both within 1e-4 of the layer's scale, are
    listed, the float64 repeat is made once more with exactly those branches taken HIP's way, and HIP's distance
    from THAT gradient must meet the same 3x-the-reference's-own-distance bound (measured on fc_example step 1,
    reconstruction phase, five entries: 5.6e-5 before, 1.1e-5 after, reference fp32 1.2e-5).
    The Adam update itself is pinned to torch.optim in tests/test_ops_gpu.py.
Free-running K-step equality is NOT tested: the trajectory is chaotic (SURVEY finding 8).
The engine runs in ``rng_mode="host"``: its tape is drawn from the global torch CPU
generator in the reference's order, which is itself part of what these tests verify.
"""
import copy
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from rankaae_amd.synthetic import make_spectra

if torch.cuda.is_available():
    from rankaae_amd import model as pm, ops
    from rankaae_amd.engine import StepEngine
    from oracle import ref_train
    DEV = torch.device("cuda:0")

KEYS = ("adversarial", "kendall", "recon", "mutual_info", "smooth")
PHASE_OF = {"adversarial": "adversarial", "correlation": "kendall", "reconstruction": "recon",
            "mutual_info": "mutual_info", "smoothness": "smooth"}


# Cases without a reference fixture: the oracle (pinned to the reference by tests/test_oracle_golden.py) is the
# expectation.  nstyle = 5 makes the first decoder block 5 -> 8 channels: not a shape the fused kernels are
# instantiated for, so their GENERIC instances (and a non-power-of-two channel count) are what runs.
INLINE_CASES = {
    "compact_nstyle5": dict(n_rows=420, n_points=256, data_seed=2, model_seed=31,
                            over=dict(ae_form="compact", nstyle=5, n_aux=3, batch_size=48)),
    # BASELINE configs[2]'s batch (4096 rows): the HBM-bound regime -- strip convolution kernel, 512-workgroup
    # grids, hundreds of gradient slabs per parameter, Adam's lane-split slab sum.
    "compact_b4096": dict(n_rows=6000, n_points=256, data_seed=4, model_seed=41,
                          over=dict(ae_form="compact", batch_size=4096)),
    "fc_b4096": dict(n_rows=6000, n_points=256, data_seed=4, model_seed=42,
                     over=dict(ae_form="FC", batch_size=4096)),
    # 1024 rows, conv networks: the large-batch kernel instances on the SERIAL chain (branches start at 1536 rows;
    # the merged backward-B + weight-gradient launches here, the per-family instance masks of raae_conv.hip)
    "compact_b1024": dict(n_rows=1600, n_points=256, data_seed=6, model_seed=43,
                          over=dict(ae_form="compact", batch_size=1024)),
}


# Ceilings on the use of P2's escape hatches (see _p2): fraction / floor of gradient tensors that may need the float64
# arbiter, tensors that may need the kink-adjusted arbiter, near-zero PReLU inputs it may account for.
P2_CEILINGS = {
    "default": dict(arbiter_frac=0.05, arbiter_min=2, kink=2, kink_entries=16),
    # measured (round 3, MI355X): fc_b4096 8 of 106 tensors through the float64 arbiter (the cancelling gradients of the
    # adversarial and smoothness phases at 4096 rows), fc_example step 1: 11 through the arbiter, 2 of them through the
    # kink-adjusted one over 5 near-zero PReLU inputs (1024 rows x 64 units x 8 layers); headroom of ~50 %
    "fc_b4096": dict(arbiter_frac=0.0, arbiter_min=12, kink=2, kink_entries=16),
    "fc_example": dict(arbiter_frac=0.0, arbiter_min=16, kink=4, kink_entries=16),
}


def load_case(case):
    if case in INLINE_CASES:
        c = INLINE_CASES[case]
        with open(os.path.join(os.path.dirname(__file__), "golden", "ref_compact_small.json")) as f:
            cfg = dict(json.load(f)["config"])
        cfg.update(c["over"])
        g = dict(config=cfg, n_rows=c["n_rows"], n_points=c["n_points"], data_seed=c["data_seed"], model_seed=c["model_seed"])
        spec, aux, _ = make_spectra(g["n_rows"], g["n_points"], cfg["n_aux"], seed=g["data_seed"])
        return g, cfg, spec, aux
    with open(os.path.join(os.path.dirname(__file__), "golden", f"ref_{case}.json")) as f:
        g = json.load(f)
    cfg = g["config"]
    spec, aux, _ = make_spectra(g["n_rows"], g["n_points"], cfg["n_aux"], seed=g["data_seed"])
    return g, cfg, spec, aux


def build_engine(cfg, seed, spec, aux, use_graph=False, rng_mode="host"):
    torch.manual_seed(seed)
    cls = pm.AE_CLS_DICT[cfg["ae_form"]]
    enc = cls["encoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], dim_in=cfg["dim_in"],
                         n_layers=cfg["n_layers"])
    dec = cls["decoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"],
                         last_layer_activation=cfg["decoder_activation"], dim_out=cfg["dim_out"],
                         n_layers=cfg["n_layers"])
    dis = pm.DiscriminatorFC(nstyle=cfg["nstyle"], dropout_rate=cfg["dis_dropout_rate"], noise=cfg["dis_noise"],
                             layers=cfg["FC_discriminator_layers"])
    eng = StepEngine(enc, dec, dis, cfg, DEV, rng_mode=rng_mode, seed=seed, use_graph=use_graph)
    n_train = ref_train.split_rows(len(spec))[0]
    eng.set_data(spec[:n_train], aux[:n_train])
    return eng


def rel_close(a, b, tol, what):
    assert abs(a - b) <= tol * abs(b) + 1e-7, f"{what}: hip {a!r} vs ref {b!r} (rel {abs(a - b) / (abs(b) + 1e-30):.2e})"


@pytest.mark.parametrize("case", ["fc_small", "fc_c2", "fc_adam_nodrop", "fc_512_aux12", "compact_small", "compact_c2",
                                  "fc_example"])
def test_p1_first_step_matches_reference_golden(case):
    g, cfg, spec, aux = load_case(case)
    torch.set_num_threads(1)
    eng = build_engine(cfg, g["model_seed"], spec, aux)
    n_train = ref_train.split_rows(len(spec))[0]
    perm = ref_train.epoch_permutation(n_train)
    alpha0 = ref_train.alpha(0.0, cfg["alpha_flat_step"], cfg["alpha_limit"])
    eng.set_epoch(perm, alpha0)
    smooth = 0 < cfg.get("epoch_stop_smooth", 500)
    eng.step(cfg["batch_size"], smooth=smooth)
    got = eng.losses()
    keys = [k for k in KEYS if not (k == "smooth" and not smooth)]
    base, bound = ref_train.derived_bounds(cfg, g["model_seed"], spec, aux, keys)
    tight, loose = ref_train.constraining(base, bound, keys)
    report = []
    for k in keys:
        if k in ("adversarial", "kendall"):     # computed before any ill-conditioned update: the reference's own value
            rel_close(got[k], g["loss_calls"][k][0], 1e-4, f"{case} step-1 {k} vs reference golden")
        err = abs(got[k] - base[k])
        report.append(f"{k}: hip {got[k]:.7g} oracle {base[k]:.7g} golden {g['loss_calls'][k][0]:.7g} "
                      f"|hip-oracle| {err:.2e} bound {bound[k]:.2e}" +
                      ("" if k in tight else f"  [bound is {bound[k] / abs(base[k]):.0%} of the value: not constraining, "
                                             "not asserted; this phase is pinned by P2 / P4]"))
        if k in tight:
            assert err <= bound[k], f"{case} step-1 {k}: |hip - oracle| = {err:.3e} exceeds the derived bound " \
                                    f"{bound[k]:.3e} (3 x the oracle's own one-ulp sensitivity + 1e-4 rel)"
    assert "adversarial" in tight and "kendall" in tight
    print(f"\n{case} P1: " + "\n  ".join(report))


def _snapshot(tr, name):
    return {"enc": {k: v.clone() for k, v in tr.encoder.state_dict().items()},
            "dec": {k: v.clone() for k, v in tr.decoder.state_dict().items()},
            "dis": {k: v.clone() for k, v in tr.discriminator.state_dict().items()},
            "opt": {id(p): {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in st.items()}
                    for p, st in tr.optimizers[name].state.items()},
            "lr": [grp["lr"] for grp in tr.optimizers[name].param_groups]}


class RandomTape:
    """Records the random tensors of one fp32 oracle step (``randn``, ``randn_like``, dropout keep-masks) and
    replays them, cast to the caller's dtype, in a second run -- so that the same step can be repeated in
    float64 as an arbiter between two fp32 implementations."""

    def __init__(self):
        import torch.nn.functional as F
        self.F, self.items, self.pos, self.replay = F, [], 0, False
        self._randn, self._randn_like, self._dropout = torch.randn, torch.randn_like, F.dropout

    def _next(self):
        v = self.items[self.pos]
        self.pos += 1
        return v

    def __enter__(self):
        def randn(*a, **k):
            if self.replay:
                return self._next().double().requires_grad_(k.get("requires_grad", False))
            v = self._randn(*a, **k)
            self.items.append(v.detach().clone())
            return v

        def randn_like(x, **k):
            if self.replay:
                return self._next().to(x.dtype)
            v = self._randn_like(x, **k)
            self.items.append(v.detach().clone())
            return v

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            if self.replay:
                return x * self._next().to(x.dtype) / (1.0 - p)
            y = self._dropout(x, p, training, False)
            self.items.append(((y != 0) | (x == 0)).detach().clone())
            return y
        torch.randn, torch.randn_like, self.F.dropout = randn, randn_like, dropout
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like, self.F.dropout = self._randn, self._randn_like, self._dropout


def prelu_modules(tr):
    """The PReLU modules of the oracle's three networks in a fixed order (the kink bookkeeping below counts their
    forward calls in execution order, which is the same in every repeat of a step)."""
    return [m for net in (tr.encoder, tr.decoder, tr.discriminator) for m in net.modules()
            if isinstance(m, torch.nn.PReLU)]


def oracle_float64_gradients(spec, aux, cfg, pre_state, post_states, tape, rows, alpha0, members, branches=None):
    """The oracle's step repeated in float64 from the same parameters, inputs and (replayed) random tensors,
    with the fp32 run's post-phase parameters forced after every phase: per phase, the exact-arithmetic
    gradients in the parameter order of the test.  ``branches = {call: [(index, positive), ...]}``: the ``call``-th
    PReLU forward of the step sees the listed input entries on the stated side of zero (where its own value lies on
    the other side a constant of twice its size is subtracted; values and gradients elsewhere are untouched) -- the
    exact gradient of the step with those branches taken."""
    state = torch.get_rng_state()
    tr = ref_train.OracleTrainer(spec, aux, cfg)
    torch.set_rng_state(state)
    mods = {"enc": tr.encoder, "dec": tr.decoder, "disc": tr.discriminator}
    for key, m in mods.items():
        m.load_state_dict(pre_state[key])
        m.double().train()
    if branches:
        calls = [0]

        def take_branches(mod, inp):
            calls[0] += 1
            todo = branches.get(calls[0] - 1)
            if todo:
                z = inp[0]
                delta = torch.zeros_like(z)
                for index, positive in todo:
                    if bool(z[index] > 0) != positive:
                        delta[index] = -2.0 * z.detach()[index]
                return (z + delta,)
        for m in prelu_modules(tr):
            m.register_forward_pre_hook(take_branches)
    grads = {}
    tr.phase_hook = lambda name: grads.__setitem__(name, [
        None if p.grad is None else p.grad.detach().clone() for grp in members[name] for p in mods[grp].parameters()])

    def force(name):
        snap = post_states[name]
        tr.encoder.load_state_dict(snap["enc"])
        tr.decoder.load_state_dict(snap["dec"])
        tr.discriminator.load_state_dict(snap["dis"])
    tr.post_hook = force
    tape.replay, tape.pos = True, 0
    with tape:
        tr.train_step(torch.tensor(tr.train_spec[rows], dtype=torch.float64),
                      torch.tensor(tr.train_aux[rows], dtype=torch.float64), alpha0, 0)
    assert tape.pos == len(tape.items)
    return grads


@pytest.mark.parametrize("case,steps", [("fc_small", (1, 2, 5, 8)), ("fc_adam_nodrop", (1, 3)), ("fc_512_aux12", (2,)),
                                        ("compact_small", (1, 2, 5, 8)), ("compact_nstyle5", (1, 3)),
                                        ("compact_b4096", (1,)), ("fc_b4096", (1,)), ("fc_example", (1, 5)),
                                        ("compact_b1024", (1,))])
def test_p2_teacher_forced_steps(case, steps):
    _p2(case, steps, use_graph=False)


@pytest.mark.parametrize("case", ["compact_b4096", "fc_b4096"])
def test_p2_teacher_forced_branched_graph_b4096(case):
    """P2 at configs[2]'s batch with ``use_graph=True``: the compared step is the REPLAY of the captured branched
    graph (the first two calls emit and capture; the hooks that force the oracle's state run between graph
    segments, so the graph is cut at the phase boundaries here and only here)."""
    _p2(case, (1,), use_graph=True)


def _p2(case, steps, use_graph):
    g, cfg, spec, aux = load_case(case)
    if use_graph and cfg["ae_form"] == "FC":
        cfg = dict(cfg, overlap_min_batch=1024)      # the dense networks' default is the serial chain: ask for the branches
    torch.set_num_threads(1)
    seed = g["model_seed"]
    # the reference's schedule: the engine otherwise defers the decoder forward that the reference runs (and discards)
    # before phase A into phase B, which is the same arithmetic but not the same state at the phase boundaries this
    # test forces (test_paired_forwards_change_nothing covers the deferred schedule bit for bit)
    eng = build_engine(dict(cfg, pair_unused_forwards=False), seed, spec, aux, use_graph=use_graph)
    torch.manual_seed(seed)
    tr = ref_train.OracleTrainer(spec, aux, cfg)
    n_train = len(tr.train_spec)
    bs = cfg["batch_size"]
    perm = ref_train.epoch_permutation(n_train)
    alpha0 = ref_train.alpha(0.3, cfg["alpha_flat_step"], cfg["alpha_limit"])   # non-zero: exercises the GRL path
    for m in (tr.encoder, tr.decoder, tr.discriminator):
        m.train()
    smooth = 0 < cfg.get("epoch_stop_smooth", 500)

    mine = {"disc": list(eng.dis_mod.parameters()), "enc": list(eng.enc_mod.parameters()),
            "dec": list(eng.dec_mod.parameters())}
    theirs = {"disc": list(tr.discriminator.parameters()), "enc": list(tr.encoder.parameters()),
              "dec": list(tr.decoder.parameters())}
    members = {"adversarial": ("disc", "enc"), "correlation": ("enc",), "reconstruction": ("enc", "dec"),
               "mutual_info": ("enc", "dec"), "smoothness": ("dec",)}
    o_params = {n: [p for grp in mem for p in theirs[grp]] for n, mem in members.items()}
    e_params = {n: [p for grp in mem for p in mine[grp]] for n, mem in members.items()}
    names_e = {id(p): pre + n for mod, pre in ((eng.dis_mod, "D."), (eng.enc_mod, "E."), (eng.dec_mod, "G."))
               for n, p in mod.named_parameters()}
    o_grads, o_post, hip_grads = {}, {}, {}
    def oracle_phase(name):
        o_grads[name] = [None if p.grad is None else p.grad.detach().clone() for p in o_params[name]]
        kink["phase_end"][name] = kink["calls"]
    tr.phase_hook = oracle_phase
    tr.post_hook = lambda name: o_post.__setitem__(name, _snapshot(tr, name))
    o_styles, hip_styles = [], {}
    tr.encoder.register_forward_hook(lambda m, i, o: o_styles.append(o.detach().clone()))
    # PReLU inputs of the oracle's dense networks, per forward call of the step (ordinal, network, layer, input):
    # where HIP's stored pre-activation lies on the other side of zero by rounding noise, the two implementations took
    # different slopes -- the kink arbiter of the gradient check below accounts for exactly those entries.
    kink = {"calls": 0, "seen": [], "phase_end": {}}
    where = {id(m): (key, i) for key, net in (("enc", tr.encoder), ("dec", tr.decoder))
             for i, m in enumerate(mm for mm in net.modules() if isinstance(mm, torch.nn.PReLU))}

    def note_kinks(mod, inp):
        if id(mod) in where and inp[0].dim() == 2:
            kink["seen"].append((kink["calls"],) + where[id(mod)] + (inp[0].detach().clone(),))
        kink["calls"] += 1
    for m in prelu_modules(tr):
        m.register_forward_pre_hook(note_kinks)

    hip_z = {}       # per phase: the pre-activations the dense networks hold when the phase's gradient is complete

    def pre(name, P):
        hip_grads[name] = eng.phase_gradient(P, name).cpu()
        hip_styles[name] = P.enc.out.detach().cpu().clone()
        if hasattr(P.enc, "z"):
            hip_z[name] = {"enc": [z.float().cpu() for z in P.enc.z], "dec": [z.float().cpu() for z in P.dec.z]}
    eng.phase_hook = pre

    def force(name, P):      # engine <- oracle state right after the oracle's optimizer step of this phase
        if name not in o_post:   # warm-up calls of the graph variant (eager emission, capture): nothing to force yet
            return
        snap = o_post[name]
        eng.enc_mod.load_state_dict(snap["enc"])
        eng.dec_mod.load_state_dict(snap["dec"])
        eng.dis_mod.load_state_dict(snap["dis"])
        o = eng.opts[name]
        for p_e, p_o in zip(e_params[name], o_params[name]):
            st = snap["opt"][id(p_o)]
            off = eng.arena.off(p_e) - o.lo
            o.m[off:off + p_e.numel()].copy_(st["exp_avg"].reshape(-1))
            o.v[off:off + p_e.numel()].copy_(st["exp_avg_sq"].reshape(-1))
    eng.post_phase_hook = force

    for k in range(1, max(steps) + 1):
        rows = perm[(k - 1) * bs:k * bs].numpy()
        if k in steps:
            def load_engine_from_oracle():
                eng.enc_mod.load_state_dict(tr.encoder.state_dict())
                eng.dec_mod.load_state_dict(tr.decoder.state_dict())
                eng.dis_mod.load_state_dict(tr.discriminator.state_dict())
                for name in members:
                    eng.load_optimizer_state(name, e_params[name], tr.optimizers[name])
            load_engine_from_oracle()
            rng_state = torch.get_rng_state()
            o_post.clear()
            if use_graph:
                # the compared call must be a REPLAY: emit eagerly and capture first (same state, same tape), ...
                for _ in range(2):
                    eng.set_epoch(perm, alpha0, start=(k - 1) * bs)
                    eng.step(len(rows), smooth=smooth)
                    torch.set_rng_state(rng_state)
                torch.cuda.synchronize()
                load_engine_from_oracle()       # ... then put the oracle's pre-step state back
                hip_grads.clear()
        o_styles.clear()
        kink.update(calls=0, seen=[], phase_end={})
        aux_b = torch.tensor(tr.train_aux[rows], dtype=torch.float32)
        arbiter = k in steps       # see the gradient check below
        if arbiter:
            pre_state = {"enc": copy.deepcopy(tr.encoder.state_dict()), "dec": copy.deepcopy(tr.decoder.state_dict()),
                         "disc": copy.deepcopy(tr.discriminator.state_dict())}
            tape = RandomTape()
            with tape:
                want = tr.train_step(torch.tensor(tr.train_spec[rows], dtype=torch.float32), aux_b, alpha0, 0)
        else:
            want = tr.train_step(torch.tensor(tr.train_spec[rows], dtype=torch.float32), aux_b, alpha0, 0)
        if k not in steps:
            continue
        after = torch.get_rng_state()
        torch.set_rng_state(rng_state)
        eng.set_epoch(perm, alpha0, start=(k - 1) * bs)
        eng.step(len(rows), smooth=smooth)
        assert torch.equal(torch.get_rng_state(), after), "host tape consumed the generator differently"
        if use_graph:
            graphs = eng.plan(len(rows)).graphs[bool(smooth)]
            assert graphs is not None and sum(isinstance(x, ops.Graph) for x in graphs) >= 5 and eng._branch
        got = eng.losses()
        bad, report = [], []
        rank_flip = False
        for key in KEYS:
            if key == "smooth" and not smooth:
                continue
            if abs(got[key] - want[key]) > 1e-4 * abs(want[key]) + 1e-7:
                if key == "kendall":     # discontinuous loss: see module docstring
                    zs_o, zs_h = o_styles[1], hip_styles["correlation"]
                    on_hip = float(ref_train.kendall_constraint(aux_b, zs_h[:, :aux_b.size(1)],
                                                                activate=cfg["kendall_activation"]))
                    if float((zs_o - zs_h).abs().max()) <= 5e-4 and abs(on_hip - got[key]) <= 1e-5 * abs(on_hip) + 1e-8:
                        rank_flip = True
                        continue
                bad.append(f"step {k} loss {key}: hip {got[key]!r} ref {want[key]!r}")
        g64 = None              # float64 repeat of this step, made on demand
        g64_kink = {}
        hatch = {"tensors": 0, "arbiter": 0, "kink": 0, "kink_entries": 0, "rank_flip": int(rank_flip)}

        def kink_adjusted(name):
            """float64 gradients of phase ``name`` with every PReLU branch HIP took differently from the oracle
            because the input is rounding noise around zero (|z| <= 1e-4 of the layer's largest input on both sides;
            the two fp32 forwards are measured 1e-6 ... 4e-5 of that scale apart, growing with depth behind the
            BatchNorms).  Per network the LAST forward of the phase is the one its gradients come from and the one
            the engine's workspace still holds.  Returns (gradients or None, [(network, layer, row, unit, z oracle,
            z hip)])."""
            ends = sorted(kink["phase_end"].values())
            first = ([0] + ends)[ends.index(kink["phase_end"][name])]
            last = {}
            for call, net, layer, z in kink["seen"]:
                if first <= call < kink["phase_end"][name]:
                    last[(net, layer)] = (call, z)
            branches, took = {}, []
            for (net, layer), (call, z) in last.items():
                h = hip_z.get(name, {}).get(net)
                if h is None or h[layer].shape != z.shape:
                    continue
                top = float(z.abs().max())
                for r, u in ((h[layer] > 0) != (z > 0)).nonzero().tolist():
                    if abs(float(z[r, u])) <= 1e-4 * top and abs(float(h[layer][r, u])) <= 1e-4 * top:
                        branches.setdefault(call, []).append(((r, u), bool(h[layer][r, u] > 0)))
                        took.append((net, layer, r, u, float(z[r, u]), float(h[layer][r, u])))
            if not took:
                return None, took
            return oracle_float64_gradients(spec, aux, cfg, pre_state, dict(o_post), tape, rows, alpha0, members,
                                            branches=branches)[name], took
        for name in members:
            if name == "smoothness" and not smooth:
                continue
            flat = hip_grads[name]
            lo = eng.opts[name].lo
            phase_max = max([float(g_.abs().max()) for g_ in o_grads[name] if g_ is not None] + [0.0])
            for ip, (p_e, g_o) in enumerate(zip(e_params[name], o_grads[name])):
                off = eng.arena.off(p_e) - lo
                mine_g = flat[off:off + p_e.numel()].view(p_e.shape).double()
                ref_g = torch.zeros_like(mine_g) if g_o is None else g_o.double()
                scale = float(ref_g.abs().max())
                err = float((mine_g - ref_g).abs().max())
                tol = 2e-2 if (rank_flip and name == "correlation") else 5e-3
                report.append((err / (scale + 1e-30), f"{name} {names_e[id(p_e)]} err {err:.2e} |g|inf {scale:.2e} "
                                                      f"phase max {phase_max:.2e} at {int((mine_g - ref_g).abs().argmax())}"
                                                      f"/{mine_g.numel()}"))
                hatch["tensors"] += 1
                if err > tol * scale + 1e-5 * phase_max + 1e-7 and g_o is not None:
                    hatch["arbiter"] += 1
                    if g64 is None:
                        g64 = oracle_float64_gradients(spec, aux, cfg, pre_state, dict(o_post), tape, rows, alpha0,
                                                       members)
                    # The reference's fp32 batch reductions (BatchNorm backward sums, GEMM accumulation) are
                    # themselves up to ~1e-2 from exact arithmetic where gradients cancel (more so at large
                    # batches), and the kernels here sum in another order / in double.  The float64 repeat of the same step is the
                    # arbiter: HIP's distance from it must be of the size of the reference's own fp32 distance (x3;
                    # measured on fc_b4096: 1.1e-5 against 6.3e-6 on a gradient of 1.5e-3).
                    e_hip = float((mine_g - g64[name][ip]).abs().max())
                    e_ref = float((ref_g - g64[name][ip]).abs().max())
                    report.append((e_hip / (scale + 1e-30), f"  ^ vs float64: hip {e_hip:.2e}, reference fp32 {e_ref:.2e}"))
                    if os.environ.get("RAAE_P2_DUMP") == f"{name}:{names_e[id(p_e)]}":
                        torch.set_printoptions(precision=6, linewidth=200, sci_mode=True)
                        print("hip - f64", (mine_g - g64[name][ip]).flatten()[:256])
                        print("ref - f64", (ref_g - g64[name][ip]).flatten()[:256])
                        print("f64", g64[name][ip].flatten()[:256])
                    if e_hip <= 3.0 * e_ref + 1e-5 * phase_max + 1e-7:
                        continue
                    # Last arbiter, for PReLU branches fp32 cannot decide: the float64 gradients of the same step
                    # with the branches HIP took at inputs that are rounding noise around zero (kink_adjusted).
                    hatch["kink"] += 1
                    if name not in g64_kink:
                        g64_kink[name] = kink_adjusted(name)
                        hatch["kink_entries"] += len(g64_kink[name][1])
                    adj, took = g64_kink[name]
                    if adj is not None and adj[ip] is not None:
                        e_hip = float((mine_g - adj[ip]).abs().max())
                        report.append((e_hip / (scale + 1e-30), f"  ^ vs float64 with HIP's PReLU branches at "
                                                                f"{len(took)} near-zero inputs: hip {e_hip:.2e}"))
                        if e_hip <= 3.0 * e_ref + 1e-5 * phase_max + 1e-7:
                            print(f"\n{case} step {k} {name} {names_e[id(p_e)]}: hip vs float64 {e_hip:.2e} (reference "
                                  f"fp32 {e_ref:.2e}) once float64 takes HIP's side at the near-zero PReLU inputs "
                                  f"(net, layer, row, unit, z oracle, z hip) {took}")
                            continue
                if err > tol * scale + 1e-5 * phase_max + 1e-7:
                    bad.append(f"step {k} {name} grad {names_e[id(p_e)]}: err {err:.3e} vs |g|inf {scale:.3e}")
        if os.environ.get("RAAE_P2_REPORT"):       # debugging aid: the largest relative gradient errors
            print(f"\n{case} step {k}: " + "\n  ".join(f"{r:.2e} {what}" for r, what in sorted(report, reverse=True)[:int(os.environ.get('RAAE_P2_REPORT'))]))
        assert not bad, f"{case}:\n" + "\n".join(bad[:40])
        # How hard the comparison had to try: the escape hatches above are each justified, but a
        # regression that pushes many tensors through them must FAIL, not pass silently.  Ceilings per case from the
        # measured counts (P2_CEILINGS) with headroom; everything else: float64 arbiter for at most 5 % of the tensors
        # (at least 2), kink-adjusted arbiter for at most 2 tensors and 16 near-zero PReLU inputs.
        ceil = dict(P2_CEILINGS.get(case, P2_CEILINGS["default"]))
        ceil_arb = max(ceil["arbiter_min"], int(ceil["arbiter_frac"] * hatch["tensors"]))
        print(f"\n{case} step {k} graph={use_graph} P2 escape hatches: {hatch['tensors']} gradient tensors, "
              f"{hatch['arbiter']} needed the float64 arbiter (ceiling {ceil_arb}), {hatch['kink']} the kink-adjusted "
              f"arbiter (ceiling {ceil['kink']}) over {hatch['kink_entries']} near-zero PReLU inputs (ceiling "
              f"{ceil['kink_entries']}), rank-loss pair flip: {bool(hatch['rank_flip'])}")
        if os.environ.get("RAAE_P2_NO_CEILING") != "1":
            assert hatch["arbiter"] <= ceil_arb, (case, k, hatch)
            assert hatch["kink"] <= ceil["kink"] and hatch["kink_entries"] <= ceil["kink_entries"], (case, k, hatch)
        # BN running statistics follow the oracle's (momentum updates of 6 enc / 4 dec forwards)
        for mod_e, mod_o in ((eng.enc_mod, tr.encoder), (eng.dec_mod, tr.decoder)):
            sd = mod_o.state_dict()
            for key, val in mod_e.state_dict().items():
                if key.endswith("running_mean") or key.endswith("running_var"):
                    assert torch.allclose(val.cpu(), sd[key], rtol=1e-4, atol=1e-6), (case, k, key)


@pytest.mark.parametrize("case", ["fc_small", "compact_small", "compact_nstyle5"])
def test_graph_replay_is_bitwise_eager(case):
    """The captured hipGraph replays the very same program: after 6 steps (3 of them replays, last
    one a ragged batch through a second plan) weights and losses are BITWISE those of eager launches,
    and two runs from the same seed are bitwise reproducible (no atomics anywhere)."""
    g, cfg, spec, aux = load_case(case)
    bs = cfg["batch_size"]
    results = []
    for use_graph in (False, True, True):
        eng = build_engine(cfg, 4321, spec, aux, use_graph=use_graph, rng_mode="philox")
        n_train = len(eng.train_spec)
        perm = torch.randperm(n_train, generator=torch.Generator().manual_seed(3))
        eng.set_epoch(perm, 0.25)
        for _ in range(5):
            eng.step(bs)
        eng.step(n_train - 5 * bs if n_train - 5 * bs < bs else bs // 2)
        torch.cuda.synchronize()
        results.append((eng.arena.P.clone(), eng.losses(), eng.opts["reconstruction"].v.clone()))
    for other in results[1:]:
        assert torch.equal(results[0][0], other[0]), "weights differ between eager and graph replay"
        assert torch.equal(results[0][2], other[2])
        assert results[0][1] == other[1]
    assert all(np.isfinite(v) for v in results[0][1].values())


@pytest.mark.parametrize("case,over", [("fc_small", {}), ("fc_example", {}), ("compact_small", {}),
                                       ("fc_b4096", {"n_steps": 3})])
def test_inline_masks_equal_tape_masks(case, over):
    """in ``rng_mode: philox`` the kernels that apply dropout regenerate their multipliers from a
    counter-based hash keyed by (seed, step, element) instead of reading a fp32 tape (``inline_masks``, default on).
    ``raae_rng_fill`` evaluates the same function for slots that stay on the tape, and every slot keeps its position in
    the numbering whether it is resident or not: with ``inline_masks: false`` (everything on the tape) the run is BIT
    FOR BIT the same -- weights, Adam moments, BatchNorm statistics, losses -- eagerly and as graph replay.  The same
    holds for the fused head of the step (``raae_step_begin``: counters, tape fill, batch gather and in-kernel spectral
    noise in one launch) against ``fused_step_begin: false`` (``raae_step_tick`` + ``raae_rng_fill`` +
    ``raae_gather_batch`` with the noise on the tape), in every combination."""
    g, cfg, spec, aux = load_case(case)
    bs = cfg["batch_size"]
    out = []
    for inline, fused in ((True, True), (False, False), (True, False), (False, True)):
        eng = build_engine(dict(cfg, inline_masks=inline, fused_step_begin=fused), 4321, spec, aux, use_graph=True,
                           rng_mode="philox")
        assert eng.inline_masks == inline and eng.fused_begin == fused
        n_train = len(eng.train_spec)
        perm = torch.randperm(n_train, generator=torch.Generator().manual_seed(3))
        eng.set_epoch(perm, 0.25)
        for _ in range(over.get("n_steps", 4)):
            if eng._host_cursor + bs > n_train:           # (the 4096-row case has one batch per epoch)
                eng.set_epoch(perm, 0.25)
            eng.step(bs)
        torch.cuda.synchronize()
        P = eng.plan(bs)
        out.append(([eng.arena.P.clone()] + [b_.clone() for mod in (eng.enc_mod, eng.dec_mod) for b_ in mod.buffers()] +
                    [o.v.clone() for o in eng.opts.values()], eng.losses(), P.tape.total))
    for other in out[1:]:
        for a, b in zip(out[0][0], other[0]):
            assert torch.equal(a, b), "in-kernel dropout multipliers / noise differ from the tape's"
        assert out[0][1] == other[1]
    assert out[0][2] < out[1][2], "multipliers and spectral noise should have left the tape"


def test_dropout_hash_statistics():
    """The counter-based hash behind the dropout multipliers (raae_common.h): over 2^22 elements of one step the
    keep fraction is keep +- 4 sigma, two steps and two seeds are uncorrelated (|corr| < 4 / sqrt(n)), and the
    multiplier is exactly 1 / keep."""
    from rankaae_amd.engine import Tape
    n, keep = 1 << 22, 0.9
    t = Tape()
    off = t.slot(n, 1, keep)
    t.finalize(DEV)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    draws = []
    for seed, step in ((7, 1), (7, 2), (8, 1)):
        ctr.fill_(step)
        ops.rng_fill(t.buf, t.seg_desc, t.seg_scale, len(t.segs), t.total, seed, ctr)
        draws.append(t.view(off, n).clone())
    sig = (keep * (1 - keep) / n) ** 0.5
    for d in draws:
        vals = torch.unique(d).tolist()
        assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - float(np.float32(1.0) / np.float32(keep))) < 1e-7
        assert abs(float((d != 0).double().mean()) - keep) < 4 * sig
    for a, b in ((0, 1), (0, 2), (1, 2)):
        x, y = (draws[a] != 0).double(), (draws[b] != 0).double()
        corr = float(((x - x.mean()) * (y - y.mean())).mean() / (x.std() * y.std()))
        assert abs(corr) < 4 / n ** 0.5, (a, b, corr)
    # neighbouring elements are uncorrelated too (lag 1, 64, 4096)
    x = (draws[0] != 0).double()
    for lag in (1, 64, 4096):
        corr = float(((x[:-lag] - x.mean()) * (x[lag:] - x.mean())).mean() / x.var())
        assert abs(corr) < 4 / n ** 0.5, (lag, corr)


def test_one_row_batch_raises_like_the_reference():
    """A last batch of ONE row: the reference's training-mode ``BatchNorm1d(nstyle)`` raises ``ValueError``
    ("Expected more than 1 value per channel when training"); so does the engine, before any launch.  Two rows
    (the smallest legal batch) step fine."""
    g, cfg, spec, aux = load_case("compact_small")
    eng = build_engine(cfg, g["model_seed"], spec, aux)
    eng.set_epoch(ref_train.epoch_permutation(len(eng.train_spec)), 0.0)
    with pytest.raises(ValueError, match="Expected more than 1 value per channel when training"):
        eng.step(1)
    eng.step(2)
    assert all(np.isfinite(v) for v in eng.losses().values())


@pytest.mark.parametrize("case,extra", [("compact_small", {}), ("fc_small", {}), ("compact_nstyle5", {}),
                                        ("compact_small", {"fused_blocks": False})])
def test_paired_forwards_change_nothing(case, extra):
    """The two forward chains whose results the reference discards run in lockstep with a needed forward chain
    (one launch per pair of block kernels; the decoder one is deferred from before phase A into phase B).  Same
    arithmetic on the same operands: after three steps every parameter, BatchNorm running statistic, Adam moment
    and loss is bit for bit what the reference's schedule gives.  Also for a generic (5-channel) block shape, and
    with ``fused_blocks: false``, where the encoder's forward has no launch to interleave with: the engine must
    then keep the reference's schedule (the deferred decoder forward would read the NEXT forward's styles)."""
    g, cfg, spec, aux = load_case(case)
    out = []
    for pair in (False, True):
        eng = build_engine(dict(cfg, pair_unused_forwards=pair, **extra), g["model_seed"], spec, aux, rng_mode="philox")
        eng.set_epoch(torch.arange(len(eng.train_spec)), 0.3)
        losses = []
        for _ in range(3):
            eng.step(cfg["batch_size"])
            losses.append(eng.losses())
        torch.cuda.synchronize()
        state = [eng.arena.P.clone()] + [b_.clone() for mod in (eng.enc_mod, eng.dec_mod) for b_ in mod.buffers()] + \
                [o.m.clone() for o in eng.opts.values()] + [o.v.clone() for o in eng.opts.values()]
        out.append((losses, state))
    assert out[0][0] == out[1][0]
    assert len(out[0][1]) == len(out[1][1]) and all(torch.equal(x, y) for x, y in zip(out[0][1], out[1][1]))


@pytest.mark.parametrize("case", ["fc_frozen", "compact_frozen", "fc_example_frozen"])
def test_p4_frozen_weights_free_running_matches_reference_golden(case, tmp_path):
    """P4 (module docstring): the product entry point -- ``Trainer.from_data(...).train()`` with hipGraph replay,
    ``rng_mode: host`` -- runs two whole epochs FREE (no teacher forcing) and every number the REAL reference
    recorded for the same seed is reproduced: 5 training losses x 16 steps (7 full batches + the ragged one per
    epoch), 5 validation losses x 2 epochs (eval-mode BatchNorm on the running statistics the training forwards
    left, trainer.py:206-268), the metrics list (Shapiro W, Spearman coupling), final running statistics, styles."""
    from rankaae_amd.parameter import Parameters
    from rankaae_amd.trainer import Trainer
    g, cfg, spec, aux = load_case(case)
    cfg = dict(cfg, rng_mode="host", use_graph=True)
    torch.set_num_threads(1)
    torch.manual_seed(g["model_seed"])

    class Quiet:
        def info(self, msg):
            pass
    tr = Trainer.from_data(None, igpu=0, verbose=False, work_dir=str(tmp_path), config_parameters=Parameters(cfg),
                           logger=Quiet(), loss_logger=Quiet(), arrays=(spec, aux))
    eng = tr.engine
    rec = {k: [] for k in KEYS}
    step0, val0 = eng.step, eng.validate

    def step(b, smooth=True):
        step0(b, smooth=smooth)
        for k, v in eng.losses().items():
            if k in rec:
                rec[k].append(v)

    def validate(*a, **kw):
        z, vl = val0(*a, **kw)
        for k in KEYS:
            rec[k].append(vl[k])
        return z, vl
    eng.step, eng.validate = step, validate
    seen = []
    tr.train(callback=lambda ep, m: seen.append([float(x) for x in m]))
    worst, bad = {}, []
    n_train, n_val, _ = ref_train.split_rows(len(spec))
    bs = cfg["batch_size"]
    rows_of_call = ([bs] * (n_train // bs) + ([n_train % bs] if n_train % bs else []) + [n_val]) * cfg["max_epoch"]
    for k in KEYS:
        want = np.array(g["loss_calls"][k])
        got = np.array(rec[k])
        assert got.shape == want.shape == (len(rows_of_call),), (k, got.shape, want.shape)
        rel = np.abs(got - want) / (np.abs(want) + 1e-30)
        worst[k] = float(rel.max())
        tol = 1e-4 * np.abs(want) + 1e-6
        if k == "kendall":
            # The rank loss weighs its concordant pairs by a ratio of pair COUNTS (functions.py:73-75).  A pair whose
            # style difference is below the fp32 distance of two implementations (styles agree to ~1e-4 behind the
            # BatchNorms of a 41-row batch) is counted on the other side: each such pair moves c_k by 4 / (b^2 - b)
            # and the loss by that times the mean concordant product / n_aux (<= 0.25 here).  Up to three such pairs
            # per call are accepted; the kernel's counts themselves are exact (tests/test_ops_gpu.py::test_rank_loss).
            b_ = np.array(rows_of_call, dtype=np.float64)
            tol = tol + 3 * 0.25 * 4.0 / (b_ * b_ - b_)
        for i in np.nonzero(np.abs(got - want) > tol)[0]:
            bad.append(f"{k} call {i} ({rows_of_call[i]} rows): hip {got[i]!r} reference {want[i]!r}")
    print(f"\n{case} P4 worst relative deviation per loss over {len(rec['recon'])} calls: "
          + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    assert not bad, f"{case}:\n" + "\n".join(bad)
    # metrics: [min Shapiro W, val recon, mean train MI, max |Spearman rho|, val rank loss] per epoch
    assert np.allclose(np.array(seen), np.array(g["epoch_metrics"]), rtol=1e-3, atol=1e-5), (seen, g["epoch_metrics"])
    eng.sync_bn_counters()
    for name, mod in (("Encoder", eng.enc_mod), ("Decoder", eng.dec_mod)):
        sd = mod.state_dict()
        for key, want in g["final_bn_buffers"][name].items():
            assert np.allclose(sd[key].double().cpu().numpy(), want, rtol=1e-4, atol=1e-6), (case, name, key)
    n_train = ref_train.split_rows(len(spec))[0]
    z, _ = eng.reconstruct(torch.tensor(spec[n_train:n_train + 8], dtype=torch.float32, device=DEV))
    assert np.allclose(z.double().cpu().numpy(), g["val_styles_first8"], rtol=1e-4, atol=1e-4)
    # weights did not move (lr 0), so the checksums are still the reference's final ones
    for name, mod in (("Encoder", eng.enc_mod), ("Decoder", eng.dec_mod), ("Style Discriminator", eng.dis_mod)):
        for key, v in mod.state_dict().items():
            if key.endswith("weight") or key.endswith("bias"):
                got = [float(v.double().sum()), float(v.double().abs().sum())]
                assert np.allclose(got, g["final_checksum"][name][key], rtol=1e-6, atol=1e-6), (name, key)


@pytest.mark.parametrize("case", ["fc_small", "compact_small", "fc_512_aux12"])
def test_validation_matches_oracle(case):
    """The per-epoch validation pass (trainer.py:206-268) on TRAINED state: the oracle trains three steps, its state
    incl. the BatchNorm running statistics goes into the engine, and ``StepEngine.validate`` must give the five
    validation losses of ``OracleTrainer.validate`` (eval-mode BatchNorm on running statistics, plain-MSE
    reconstruction, rank loss over the n_val^2 pairs, mutual information with randn[n_val, nstyle], adversarial
    without input noise / dropout but with z_real of the CONFIGURED batch size) to rel 1e-4 and its styles to 1e-5,
    consuming the global generator identically; eager, captured and replayed."""
    g, cfg, spec, aux = load_case(case)
    torch.set_num_threads(1)
    eng = build_engine(cfg, g["model_seed"], spec, aux, use_graph=True)
    torch.manual_seed(g["model_seed"])
    tr = ref_train.OracleTrainer(spec, aux, cfg)
    for m in (tr.encoder, tr.decoder, tr.discriminator):
        m.train()
    bs = cfg["batch_size"]
    perm = ref_train.epoch_permutation(len(tr.train_spec)).numpy()
    alpha_ = ref_train.alpha(0.4, cfg["alpha_flat_step"], cfg["alpha_limit"])
    for k in range(3):
        rows = perm[k * bs:(k + 1) * bs]
        tr.train_step(torch.tensor(tr.train_spec[rows], dtype=torch.float32),
                      torch.tensor(tr.train_aux[rows], dtype=torch.float32), alpha_, 0)
    eng.enc_mod.load_state_dict(tr.encoder.state_dict())
    eng.dec_mod.load_state_dict(tr.decoder.state_dict())
    eng.dis_mod.load_state_dict(tr.discriminator.state_dict())
    val_spec, val_aux = tr.val_spec.to(DEV), tr.val_aux.to(DEV)
    eng.alpha_dev.fill_(float(alpha_))
    for rep in range(3):                      # eager, capture + launch, replay
        state = torch.get_rng_state()
        z_o, want = tr.validate(alpha_)
        after = torch.get_rng_state()
        torch.set_rng_state(state)
        z_h, got = eng.validate(val_spec, val_aux)
        assert torch.equal(torch.get_rng_state(), after), "validation consumed the generator differently"
        for k in KEYS:
            rel_close(got[k], want[k], 1e-4, f"{case} validation {k} (call {rep})")
        assert torch.allclose(z_h.cpu(), z_o.detach(), rtol=1e-5, atol=1e-5), float((z_h.cpu() - z_o).abs().max())
        W, rho = eng.val_style_metrics()
        from scipy.stats import shapiro
        assert np.allclose(W, [shapiro(x).statistic for x in z_h.cpu().numpy().T], rtol=0, atol=1e-12)


@pytest.mark.parametrize("ae_form,depth", [("compact", 2), ("FC", 2), ("compact", 1)])
def test_branched_graph_is_bitwise_eager_b4096(ae_form, depth):
    """BASELINE configs[2]'s batch: from ``overlap_min_batch`` rows up the captured step is a BRANCHED graph (weight
    gradients on side streams, discarded forwards on the auxiliary stream).  Replaying it must be bit for bit the
    eager launches, and two captures must agree with each other.  ``depth`` = 2 (the default since the end of round 3):
    two forked batches of weight-gradient launches in flight at once (``wgrad_overlap_depth``) -- deeper overlap is
    where round 1 once saw a captured graph differ from eager launches; with today's kernels it does not.  The engine caps the
    depth at 2: a step graph with FIVE concurrent branches (depth 3: main chain, auxiliary stream, three side
    streams) crashed inside ``hipGraphLaunch`` on ROCm 7.0 / 7.2 in long test sessions (nets_conv.py, fork_wgrad)."""
    g, cfg, spec, aux = load_case("compact_b4096" if ae_form == "compact" else "fc_b4096")
    cfg = dict(cfg, wgrad_overlap_depth=depth)
    if ae_form == "FC":
        cfg["overlap_min_batch"] = 1024              # the dense networks' default is the serial chain: ask for the branches
    results = []
    for use_graph in (False, True, True):
        eng = build_engine(cfg, 777, spec, aux, use_graph=use_graph, rng_mode="philox")
        assert cfg["batch_size"] >= eng.overlap_min_batch
        eng.set_epoch(torch.randperm(len(eng.train_spec), generator=torch.Generator().manual_seed(5)), 0.3)
        for _ in range(5):
            eng.set_epoch(eng.perm.cpu(), 0.3)          # 4200 training rows: one 4096-row batch per epoch
            eng.step(4096)
        assert eng._branch
        torch.cuda.synchronize()
        results.append((eng.arena.P.clone(), eng.losses(), eng.opts["reconstruction"].v.clone(),
                        [b_.clone() for mod in (eng.enc_mod, eng.dec_mod) for b_ in mod.buffers()]))
    for other in results[1:]:
        assert torch.equal(results[0][0], other[0]), "weights differ between eager and branched-graph replay"
        assert torch.equal(results[0][2], other[2])
        assert results[0][1] == other[1]
        assert all(torch.equal(x, y) for x, y in zip(results[0][3], other[3]))
    assert all(np.isfinite(v) for v in results[0][1].values())


def test_bf16_storage_mode_fc_512_aux12():
    """BASELINE configs[4] (512-point spectra, 12 descriptors, nstyle 13, dense networks) with the build-only key
    ``precision: bf16``: hidden activations and dropout multipliers are STORED as bf16, every product, sum,
    BatchNorm statistic, gradient and Adam moment stays fp32 / double.  The reference arithmetic is fp32
    (sc/clustering/dataloader.py:61) and the 1e-4 parity gate applies to the default fp32 mode only; for bf16 the
    stated bound is: teacher-forced from the same state and random tape, every phase loss within 2e-2 relative of
    the fp32 engine's (the rank loss, a cancelling sum: 5e-3 absolute) (bf16 keeps 8 significant bits, 4e-3 per element; five dense layers and a BatchNorm each way),
    the phase gradients are REPORTED and only loosely bounded (relative L2 error <= 0.4).
    Measured: rank phase 1.6 %, adversarial 12 %, smoothness 8 %, reconstruction / mutual information 20-25 %.  The
    cause is the architecture, not the kernels: BatchNorm1d(affine=False) behind PReLU(0.01) normalises "dead" units
    (all-negative pre-activations, batch std ~1e-4 after the slope) back to unit variance, so a bf16 step on the stored
    pre-activation (2^-8 relative) comes out of the BatchNorm as O(0.1) noise; fp32 storage keeps 16 more bits there.
    The first layer's output is kept fp32 for the same reason (its input, the raw spectrum, is not normalised: with
    it in bf16 the errors were 40-70 %).  DESIGN.md section 7 says so: bf16 storage is offered because BASELINE
    configs[4] names it, fp32 stays the default and the only mode the 1e-4 parity gate applies to.  The fp32 mode itself must not notice that the bf16 mode exists: two
    fp32 engines from the same seed stay bit for bit equal, and an fp32 engine never passes a storage bit."""
    g, cfg, spec, aux = load_case("fc_512_aux12")
    bs = cfg["batch_size"]
    runs = {}
    for prec in ("fp32", "fp32", "bf16"):
        torch.manual_seed(99)
        eng = build_engine(dict(cfg, precision=prec, pair_unused_forwards=False), g["model_seed"], spec, aux)
        assert eng.bf16 == (prec == "bf16")
        assert all(z.dtype == (torch.bfloat16 if prec == "bf16" and 0 < i < 4 else torch.float32)
                   for i, z in enumerate(eng.plan(bs).enc.z))
        grads, losses = {}, []
        eng.phase_hook = lambda name, P: grads.__setitem__(name, eng.phase_gradient(P, name).cpu())
        state0 = eng.arena.P.clone()

        def reset(name, P):          # teacher forcing: every phase starts from the initial weights
            eng.arena.P.copy_(state0)
        eng.post_phase_hook = reset
        torch.manual_seed(5)         # the host tape: same draws for every run
        eng.set_epoch(torch.arange(len(eng.train_spec)), 0.3)
        eng.step(bs)
        runs.setdefault(prec, []).append((eng.losses(), grads))
    a, b = runs["fp32"]
    assert a[0] == b[0] and all(torch.equal(a[1][k], b[1][k]) for k in a[1]), "fp32 mode is not reproducible bit for bit"
    (l16, g16), (l32, g32) = runs["bf16"][0], a
    assert l16 != l32, "bf16 storage changed nothing: the mode is not active"
    for k in KEYS:
        # (the rank loss is a signed sum of O(0.5) pair products that nearly cancel -- 0.024 here: its bound is absolute,
        # the 4e-3 a bf16-rounded style moves each product by)
        tol = 5e-3 if k == "kendall" else 2e-2 * abs(l32[k]) + 1e-6
        assert abs(l16[k] - l32[k]) <= tol, (k, l16[k], l32[k])
    report = {}
    for name in g32:
        err, scale = float((g16[name] - g32[name]).abs().max()), float(g32[name].abs().max())
        l2 = float((g16[name] - g32[name]).norm() / g32[name].norm())
        report[name] = (round(err / scale, 4), round(l2, 4))
    print("\nbf16 vs fp32 phase gradients (max error / |g|inf, relative L2 error):", report)
    for name, (emax, el2) in report.items():
        assert el2 <= 0.4 and emax <= 0.45, (name, emax, el2)
    print("\nbf16 vs fp32 losses:", {k: (round(l16[k], 6), round(l32[k], 6)) for k in KEYS})
    # free-running smoke: hipGraph replay with the device tape, finite losses after 6 steps
    eng = build_engine(dict(cfg, precision="bf16"), g["model_seed"], spec, aux, use_graph=True, rng_mode="philox")
    eng.set_epoch(torch.arange(len(eng.train_spec)), 0.3)
    for _ in range(3):
        eng.step(bs)
    assert all(np.isfinite(v) for v in eng.losses().values())
    with pytest.raises(ValueError, match="precision: bf16"):
        build_engine(dict(load_case("compact_small")[1], precision="bf16"), 1, *load_case("compact_small")[2:])


@pytest.mark.parametrize("case", ["fc_small", "compact_small", "compact_nstyle5", "fc_example"])
def test_overlapped_steps_are_bitwise_the_plain_steps(case):
    """`overlap_steps` (experiment, off by default): the decoder-only rest of a step's smoothness phase runs on a second
    stream beside the NEXT step's phase A (encoder and discriminator only).  Same kernels, same operands, same order wherever one depends on another: after two epochs -- full batches,
    a ragged one (another plan: the pending tail is run first), steps without the smoothness phase, a validation in
    between and a read of the losses (both complete the pending tail) -- parameters, Adam moments, BatchNorm
    statistics and every loss are BIT FOR BIT those of the engine that runs each step whole."""
    g, cfg, spec, aux = load_case(case)
    bs = cfg["batch_size"]
    n_train, n_val = ref_train.split_rows(len(spec))[:2]
    ragged = n_train - 3 * bs if 2 <= n_train - 3 * bs < bs else bs // 2
    vs = torch.tensor(spec[n_train:n_train + n_val], dtype=torch.float32, device=DEV)
    va = torch.tensor(aux[n_train:n_train + n_val], dtype=torch.float32, device=DEV)

    def run(overlap):
        e = build_engine(dict(cfg, overlap_steps=overlap), 77, spec, aux, use_graph=True, rng_mode="philox")
        assert e.defer_tail == overlap
        out = []
        for ep in range(3):
            e.set_epoch(torch.randperm(n_train, generator=torch.Generator().manual_seed(10 + ep)), 0.4)
            for k in range(3):
                e.step(bs, smooth=ep < 2)
                if ep == 1 and k == 1:
                    out.append(e.losses())          # completes the pending tail in the middle of an epoch
            e.step(ragged, smooth=ep < 2)
            z, vl = e.validate(vs, va)
            out.append((e.losses(), vl, z.clone()))
        torch.cuda.synchronize()
        state = ([e.arena.P.clone()] + [b_.clone() for mod in (e.enc_mod, e.dec_mod) for b_ in mod.buffers()] +
                 [o.m.clone() for o in e.opts.values()] + [o.v.clone() for o in e.opts.values()] + [e.steps_dev.clone()])
        pending = e._tail
        e.release()
        return out, state, pending
    plain, sp, _ = run(False)
    over, so, pending = run(True)
    assert pending is None
    for a, b in zip(plain, over):
        if isinstance(a, dict):
            assert a == b
        else:
            assert a[0] == b[0] and a[1] == b[1] and torch.equal(a[2], b[2])
    for a, b in zip(sp, so):
        assert torch.equal(a, b)


def test_tile_hint_changes_only_the_rounding():
    """``tile_rows_mult`` (raae_tile_hint) regroups the samples of the conv-network launches: other partial sums, the
    same arithmetic.  With the weights frozen (lr_base = 0, as in P4) four free-running steps of the same seed give the
    five losses of every step within 1e-5 relative of the default geometry's and the BatchNorm running statistics
    within 1e-5 (the partial statistics are float64: usually every bit agrees) from fewer workgroups; and the hint is per engine -- an engine without it, run afterwards on the same thread, is bitwise
    what it was before."""
    g, cfg, _, _ = load_case("compact_small")
    cfg = dict(cfg, lr_base=0.0, batch_size=256)
    spec, aux, _ = make_spectra(1600, g["n_points"], cfg["n_aux"], seed=9)
    n_train = ref_train.split_rows(len(spec))[0]

    def run(mult):
        c = dict(cfg) if mult is None else dict(cfg, tile_rows_mult=mult)
        e = build_engine(c, 77, spec, aux, use_graph=True, rng_mode="philox")
        e.set_epoch(torch.randperm(n_train, generator=torch.Generator().manual_seed(3)), 0.5)
        out = []
        for _ in range(4):
            e.step(256)
            out.append(e.losses())
        torch.cuda.synchronize()
        stats = [b_.clone() for mod in (e.enc_mod, e.dec_mod) for b_ in mod.buffers()]
        parts = [w.nY for w in e.plans[256].enc.blk] + [w.nY for w in e.plans[256].dec.blk]   # partial-statistic rows = workgroups
        e.release()
        return (out, parts), stats
    base, sb = run(None)
    hint, sh = run(4)
    again, sa = run(None)
    assert base == again and all(torch.equal(x, y) for x, y in zip(sb, sa))
    assert sum(hint[1]) < sum(base[1]), ("the hint did not change the launch geometry", base[1], hint[1])
    for l0, l1 in zip(base[0], hint[0]):
        for k in KEYS:
            # (the rank loss weighs pair COUNTS: a pair whose style difference is rounding noise moves it by 4/(b^2-b))
            tol = 3 * 4.0 / (256 * 255) if k == "kendall" else 1e-5 * max(abs(l0[k]), 1e-3)
            assert abs(l0[k] - l1[k]) <= tol, (k, l0[k], l1[k])
    for x, y in zip(sb, sh):
        if x.dtype.is_floating_point:
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("case,T", [("fc_small", 3), ("fc_example", 2), ("compact_small", 3), ("compact_nstyle5", 2),
                                    ("compact_small+tile4", 3)])
def test_trial_batch_is_bitwise_the_trials_alone(case, T):
    """VERDICT r2 item 3 / SURVEY 8f-3: T independent trials of the dense networks stepped by ONE launch sequence whose
    kernels run with gridDim.z = T (``rankaae_amd.trial_batch.TrialBatch``).  Plane z runs the body the trial runs
    alone on the trial's own buffers: after six steps (eager first step, captured graph, replays, a ragged batch
    through a second program) every trial's weights, Adam moments, BatchNorm statistics and losses are BIT FOR BIT those
    of the same trial stepped by itself."""
    from rankaae_amd.trial_batch import TrialBatch
    g, cfg, spec, aux = load_case(case.split("+")[0])
    if case.endswith("+tile4"):          # train_sc's batched mode for the conv networks: launch geometry of a 4x larger batch
        cfg = dict(cfg, tile_rows_mult=4)
    bs = cfg["batch_size"]
    n_train = ref_train.split_rows(len(spec))[0]
    ragged = n_train - 3 * bs if 2 <= n_train - 3 * bs < bs else bs // 2

    def make(t, stream=None):
        torch.manual_seed(100 + t)
        cls = pm.AE_CLS_DICT[cfg["ae_form"]]
        enc = cls["encoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], dim_in=cfg["dim_in"], n_layers=cfg["n_layers"])
        dec = cls["decoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], last_layer_activation=cfg["decoder_activation"],
                             dim_out=cfg["dim_out"], n_layers=cfg["n_layers"])
        dis = pm.DiscriminatorFC(nstyle=cfg["nstyle"], dropout_rate=cfg["dis_dropout_rate"], noise=cfg["dis_noise"],
                                 layers=cfg["FC_discriminator_layers"])
        eng = StepEngine(enc, dec, dis, cfg, DEV, rng_mode="philox", seed=500 + t, use_graph=True, stream=stream)
        eng.set_data(spec[:n_train], aux[:n_train])
        return eng

    def state(e):
        torch.cuda.synchronize()
        return ([e.arena.P.clone()] + [b_.clone() for mod in (e.enc_mod, e.dec_mod) for b_ in mod.buffers()] +
                [o.m.clone() for o in e.opts.values()] + [o.v.clone() for o in e.opts.values()], e.losses())
    n_val = ref_train.split_rows(len(spec))[1]
    vs = torch.tensor(spec[n_train:n_train + n_val], dtype=torch.float32, device=DEV)
    va = torch.tensor(aux[n_train:n_train + n_val], dtype=torch.float32, device=DEV)
    # a split of >= 1024 rows: the conv networks' large-batch kernel instances have no batched form, and the batch then
    # runs the trials' validations one after the other inside its graph
    reps = -(-1100 // n_val)
    vs_big, va_big = vs.repeat(reps, 1)[:1100].contiguous(), va.repeat(reps, 1)[:1100].contiguous()

    def perm(t, ep):
        return torch.randperm(n_train, generator=torch.Generator().manual_seed(1000 * t + ep))
    alone = []
    for t in range(T):
        e = make(t)
        for ep in range(2):
            e.set_epoch(perm(t, ep), 0.3)
            for _ in range(3):
                e.step(bs)
            e.step(ragged)
        vals = []
        for x, y in ((vs, va), (vs_big, va_big)):
            for _ in range(3):                     # eager, captured, replayed
                z, vl = e.validate(x, y)
                vals.append((z.clone(), vl, [m.copy() for m in e.val_style_metrics()]))
        alone.append(state(e) + (vals,))
    shared = TrialBatch.shared_stream(DEV)
    engs = [make(t, shared) for t in range(T)]
    batch = TrialBatch(engs)
    for ep in range(2):
        for t, e in enumerate(engs):
            e.set_epoch(perm(t, ep), 0.3)
        for _ in range(3):
            batch.step(bs)
        batch.step(ragged)
    assert batch.programs[(bs, True)][1] is not None and batch.launches_per_step(bs) > 50
    for t, e in enumerate(engs):
        got = state(e)
        for a, b in zip(alone[t][0], got[0]):
            assert torch.equal(a, b), f"trial {t} differs from the same trial alone"
        assert alone[t][1] == got[1]
    # the per-epoch validation as one launch sequence: eager + logged, captured, replayed -- the alone engine's numbers
    for rep in range(6):
        x, y = (vs, va) if rep < 3 else (vs_big, va_big)
        res = batch.validate([x] * T, [y] * T)
        for t, e in enumerate(engs):
            z0, vl0, met0 = alone[t][2][rep]
            assert torch.equal(res[t][0], z0) and res[t][1] == vl0, (t, rep, res[t][1], vl0)
            W, rho = e.val_style_metrics()
            assert (W == met0[0]).all() and (rho == met0[1]).all()
    batch.release()
