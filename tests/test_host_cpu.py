"""CPU tests of the host-side mirror of the reference interface (no GPU, no HIP compute):
configuration object, dataset split / shuffle order, seed parity of the parameter containers,
LR schedule, data-parallel sharding over gloo (world_size 2)."""
import json
import os

import numpy as np
import pytest
import torch

from rankaae_amd import model as pm
from rankaae_amd.dataloader import get_dataloaders, split_counts
from rankaae_amd.parameter import Parameters
from rankaae_amd.synthetic import make_spectra, write_csv
from rankaae_amd.trainer import PlateauScheduler, alpha

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_parameters_mirror_reference_semantics(tmp_path):
    """Same behaviour the reference pins in sc/tests/test_parameters.py."""
    d = dict(nstyle=2, weight_decay=1e-2, lr_ratio_Reconn=2.0, optimizer_name="AdamW", aux_weights=None,
             kendall_activation=False)
    p = Parameters(d)
    assert p.get("nstyle", 0) == 2 and p.get("nstyll", 0) == 0
    with pytest.raises(TypeError):
        p.nstyle = 3
    p.update({"new_parameter1": 1.2})
    assert p.new_parameter1 == 1.2 and p.nstyle == 2
    d.update({"nstyle": 3, "kendall_activation": True})
    p.update(d)
    assert p.nstyle == 3 and p.kendall_activation is True and p.to_dict()["nstyle"] == 3
    y = tmp_path / "fix_config.yaml"
    y.write_text("ae_form: FC\nalpha_limit: 0.7172\nn_aux: 5\n")
    q = Parameters.from_yaml(str(y))
    assert q.ae_form == "FC" and q.alpha_limit == 0.7172


@pytest.mark.parametrize("case", ["fc_small", "compact_small", "fc_512_aux12"])
def test_containers_reproduce_reference_initial_weights(case):
    """Constructing the product's containers in the reference's order from the same seed gives the
    reference's initial weights (checksums stored in the golden fixtures) and state_dict keys."""
    with open(os.path.join(GOLDEN, f"ref_{case}.json")) as f:
        g = json.load(f)
    c = g["config"]
    torch.manual_seed(g["model_seed"])
    cls = pm.AE_CLS_DICT[c["ae_form"]]
    enc = cls["encoder"](nstyle=c["nstyle"], dropout_rate=c["dropout_rate"], dim_in=c["dim_in"], n_layers=c["n_layers"])
    dec = cls["decoder"](nstyle=c["nstyle"], dropout_rate=c["dropout_rate"], last_layer_activation=c["decoder_activation"],
                         dim_out=c["dim_out"], n_layers=c["n_layers"])
    dis = pm.DiscriminatorFC(nstyle=c["nstyle"], dropout_rate=c["dis_dropout_rate"], noise=c["dis_noise"],
                             layers=c["FC_discriminator_layers"])
    for name, mod in (("Encoder", enc), ("Decoder", dec), ("Style Discriminator", dis)):
        want = g["init_checksum"][name]
        sd = mod.state_dict()
        assert list(sd.keys()) == list(want.keys())
        for k, v in sd.items():
            got = [float(v.double().sum()), float(v.double().abs().sum())]
            assert np.allclose(got, want[k], rtol=1e-12, atol=1e-12), (name, k)


def test_dataset_split_and_shuffle_order(tmp_path):
    spec, aux, grid = make_spectra(100, 16, 5, seed=3)
    csv = tmp_path / "d.csv"
    write_csv(str(csv), spec, aux, grid)
    tr, va, te = get_dataloaders(str(csv), 8, (0.7, 0.15, 0.15), n_aux=5)
    assert [len(x.dataset) for x in (tr, va, te)] == split_counts(100) == [70, 15, 15]
    assert len(tr) == 9                      # last, partial batch is kept (drop_last=False)
    np.testing.assert_allclose(tr.dataset.spec, spec[:70])
    np.testing.assert_allclose(va.dataset.aux, aux[70:85])
    assert split_counts(700)[0] == 489       # int(700 * 0.7) as the reference computes it
    from oracle.ref_train import epoch_permutation
    torch.manual_seed(5)
    a = tr.epoch_permutation()
    torch.manual_seed(5)
    b = epoch_permutation(70)
    assert torch.equal(a, b)
    torch.manual_seed(5)
    rows = [x[0].shape[0] for x in tr]
    assert rows == [8] * 8 + [6]


def test_plateau_scheduler_matches_torch():
    class Opt:
        lr = 0.01

        def push(self):
            pass
    mine = PlateauScheduler(Opt(), factor=0.1, patience=3)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=0.01)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.1, patience=3, cooldown=0, threshold=0.01)
    rng = np.random.default_rng(0)
    for m in np.concatenate([np.linspace(1, 0.5, 10), 0.5 + 0.001 * rng.standard_normal(30), -0.2 - 0.01 * np.arange(10)]):
        mine.step(m)
        ref.step(m)
        assert abs(mine.opt.lr - opt.param_groups[0]["lr"]) < 1e-15


def test_alpha_ramp():
    assert alpha(0.0, 739, 0.7172) == 0.0
    assert abs(alpha(1.0, 739, 0.7172) - 0.7172) < 1e-5


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rankaae_amd.engine import StepEngine
    with pytest.raises(RuntimeError, match="MI355X"):
        StepEngine(None, None, None, {}, torch.device("cpu"))


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    from rankaae_amd.parallel import allreduce_mean_, cursor_params, full_global_batches, shard_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    perm = torch.randperm(100, generator=torch.Generator().manual_seed(1))
    b = 8
    nb = full_global_batches(100, world, b)
    start, stride = cursor_params(rank, world, b)
    mine = [shard_rows(perm, rank, world, b, i) for i in range(nb)]
    for i, rows in enumerate(mine):            # what the device cursor walks: perm[start + i*stride : +b]
        assert torch.equal(rows, perm[start + i * stride:start + i * stride + b])
    gathered = [None] * world
    dist.all_gather_object(gathered, torch.cat(mine).tolist())
    flat = sum(gathered, [])
    ok_cover = sorted(flat) == sorted(perm[:nb * world * b].tolist()) and len(set(flat)) == len(flat)
    g = torch.full((64,), float(rank + 1))
    allreduce_mean_(g)
    ok_mean = bool(torch.allclose(g, torch.full((64,), (world + 1) / 2.0)))
    # the stop flag of a data-parallel run (Trainer.request_stop): raised on ONE rank, seen by all
    from rankaae_amd.parallel import any_rank
    ok_mean = ok_mean and any_rank(rank == 1, torch.device("cpu")) and not any_rank(False, torch.device("cpu"))
    if rank == 0:
        out.put((nb, ok_cover, ok_mean))
    dist.destroy_process_group()


def test_data_parallel_sharding_and_gradient_mean_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    nb, ok_cover, ok_mean = out.get(timeout=10)
    assert nb == 6 and ok_cover and ok_mean


@pytest.mark.parametrize("n", [3, 4, 5, 6, 11, 36, 1050, 1051, 4999])
def test_shapiro_coefficients_reproduce_scipy_W(n):
    """The host-built coefficient vector of ``raae_style_metrics`` (AS R94 + AS 111) is scipy's: W formed from
    it with AS R94's arithmetic equals ``scipy.stats.shapiro(x).statistic`` (public API) to rounding."""
    from scipy.stats import shapiro
    from rankaae_amd.metrics import shapiro_coefficients
    a = shapiro_coefficients(n)
    assert a.shape == (n // 2,)
    rng = np.random.default_rng(n)
    for x in (rng.standard_normal(n).astype(np.float32), rng.exponential(size=n).astype(np.float32)):
        y = np.sort(x.astype(np.float64)) - np.float64(x[n // 2])
        coef = np.zeros(n)
        coef[:n // 2] = -a
        coef[n - n // 2:] = a[::-1]
        u = y / (y[-1] - y[0])
        asa, xsx = coef - coef.sum() / n, u - u.sum() / n
        ssa, ssx, sax = (asa * asa).sum(), (xsx * xsx).sum(), (asa * xsx).sum()
        s = np.sqrt(ssa * ssx)
        w = 1.0 - (s - sax) * (s + sax) / (ssa * ssx)
        assert abs(w - shapiro(x).statistic) < 1e-13, (n, w, shapiro(x).statistic)


def test_csv_binary_cache_is_transparent(tmp_path):
    """the second load of a CSV comes from ``<csv>.raae_cache.npz`` with identical arrays and
    index; touching the CSV invalidates it."""
    from rankaae_amd import dataloader as dl
    spec, aux, grid = make_spectra(60, 32, 3, seed=3)
    csv = str(tmp_path / "d.csv")
    write_csv(csv, spec, aux, grid)
    a = dl.load_csv(csv, 3)
    assert os.path.exists(csv + dl.CACHE_SUFFIX)
    calls = []
    orig = dl._parse_csv
    dl._parse_csv = lambda *k: (calls.append(1), orig(*k))[1]
    try:
        b = dl.load_csv(csv, 3)
        assert not calls                                   # served from the cache
        ref = orig(csv, 3)
        for x, y, r in zip(a[:3], b[:3], ref[:3]):
            assert x.dtype == y.dtype == r.dtype == np.float64
            assert np.array_equal(x, y) and np.array_equal(x, r)
        assert list(a[3]) == list(b[3]) == list(ref[3])
        dl.load_csv(csv, 3, cache=False)
        assert len(calls) == 1
        os.utime(csv, ns=(1, 1))                           # a changed CSV must be parsed again
        dl.load_csv(csv, 3)
        assert len(calls) == 2
    finally:
        dl._parse_csv = orig


def test_trial_assignment_round_robin():
    from rankaae_amd.cmd.train_sc import assign_trials
    assert assign_trials(5, 2) == [[0, 2, 4], [1, 3]]
    assert assign_trials(3, 8)[:3] == [[0], [1], [2]] and all(not j for j in assign_trials(3, 8)[3:])
    assert sorted(k for part in assign_trials(32, 8) for k in part) == list(range(32))


@pytest.mark.parametrize("case", ["fc_frozen", "compact_frozen"])
def test_exported_modules_forward_is_the_reference_forward(case):
    """``final.pt`` consumers (the reference's report tool, ``sc/report/generate_report.py:272-275``) call the
    plain-PyTorch ``forward`` of ``rankaae_amd.model``'s containers.  Pin that arithmetic (reference
    ``sc/clustering/model.py:330-378, 264-295, 430-474, 518-570, 631-663``):
      * with the oracle's ``state_dict`` loaded, train-mode (same dropout draws, BatchNorm batch statistics and
        running-statistics updates) and eval-mode forwards of encoder, decoder and discriminator equal
        ``oracle.ref_model``'s to 1e-6;
      * through the oracle's trained state of a ``*_frozen`` fixture (lr_base = 0: the weights stay at their
        initial values, the BatchNorm running statistics of two epochs are not chaotic -- see oracle/gen_golden.py)
        the eval-mode encoder reproduces the REFERENCE's ``val_styles_first8`` to 1e-5."""
    from oracle import ref_train
    with open(os.path.join(GOLDEN, f"ref_{case}.json")) as f:
        g = json.load(f)
    c = g["config"]
    spec, aux, _ = make_spectra(g["n_rows"], g["n_points"], c["n_aux"], seed=g["data_seed"])
    torch.set_num_threads(1)
    torch.manual_seed(g["model_seed"])
    tr = ref_train.OracleTrainer(spec, aux, c)
    tr.train()                                                  # 2 epochs; running statistics move, weights do not
    cls = pm.AE_CLS_DICT[c["ae_form"]]
    enc = cls["encoder"](nstyle=c["nstyle"], dropout_rate=c["dropout_rate"], dim_in=c["dim_in"], n_layers=c["n_layers"])
    dec = cls["decoder"](nstyle=c["nstyle"], dropout_rate=c["dropout_rate"], last_layer_activation=c["decoder_activation"],
                         dim_out=c["dim_out"], n_layers=c["n_layers"])
    dis = pm.DiscriminatorFC(nstyle=c["nstyle"], dropout_rate=c["dis_dropout_rate"], noise=c["dis_noise"],
                             layers=c["FC_discriminator_layers"])
    for mine, theirs in ((enc, tr.encoder), (dec, tr.decoder), (dis, tr.discriminator)):
        mine.load_state_dict(theirs.state_dict(), strict=True)
    n_train = int(g["n_rows"] * 0.7)
    x = torch.tensor(spec[n_train:n_train + 48], dtype=torch.float32)
    z_in = torch.randn(48, c["nstyle"])
    for train in (True, False):
        for m in (enc, dec, dis, tr.encoder, tr.decoder, tr.discriminator):
            m.train(train)
        outs = []
        for e_, d_, s_ in ((enc, dec, dis), (tr.encoder, tr.decoder, tr.discriminator)):
            torch.manual_seed(99)                               # same dropout / input-noise draws on both sides
            with torch.no_grad():
                z = e_(x)
                outs.append((z, d_(z), d_(z_in), s_(z, 0.3)))
        for a, b in zip(*outs):
            assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-6, atol=1e-6), (case, train)
        if train:                                               # the train-mode forwards moved the running statistics alike
            for mine, theirs in ((enc, tr.encoder), (dec, tr.decoder)):
                for (k, v), (_, w) in zip(mine.state_dict().items(), theirs.state_dict().items()):
                    assert torch.allclose(v.double(), w.double(), rtol=1e-6, atol=1e-7), (case, k)
    # the gradient-reversal layer of the container (model.py:8-22)
    zz = torch.randn(5, c["nstyle"], requires_grad=True)
    dis.eval()
    dis(zz, 0.25).sum().backward()
    zz2 = zz.detach().clone().requires_grad_(True)
    tr.discriminator.eval()
    tr.discriminator(zz2, 0.25).sum().backward()
    assert torch.allclose(zz.grad, zz2.grad, rtol=1e-6, atol=1e-7)
    # ... and against the reference itself: reload the post-training state (the train-mode forwards above moved it)
    torch.manual_seed(g["model_seed"])
    tr2 = ref_train.OracleTrainer(spec, aux, c)
    tr2.train()
    enc.load_state_dict(tr2.encoder.state_dict())
    enc.eval()
    with torch.no_grad():
        st = enc(torch.tensor(spec[n_train:n_train + 8], dtype=torch.float32)).double().numpy()
    assert np.allclose(st, g["val_styles_first8"], rtol=1e-5, atol=1e-5), np.abs(st - np.array(g["val_styles_first8"])).max()
    for key, want in g["final_bn_buffers"]["Encoder"].items():
        assert np.allclose(enc.state_dict()[key].double().numpy(), want, rtol=1e-5, atol=1e-6), key


def test_data_parallel_epoch_schedule_covers_the_permutation():
    """``parallel.epoch_schedule``: full global batches then the tail split evenly over the ranks; the ranks' row
    ranges are disjoint, in order, and leave at most world - 1 rows unused; world = 1 is the reference's loader
    (every batch, the ragged last one included)."""
    from rankaae_amd.parallel import epoch_schedule
    assert epoch_schedule(489, 1, 64) == [(64, 64 * i, 64) for i in range(7)] + [(41, 448, 41)]
    assert epoch_schedule(128, 1, 64) == [(64, 0, 64), (64, 64, 64)]
    for n, w, b in ((489, 2, 32), (4900, 8, 256), (4900, 4, 256), (70, 2, 64), (10, 4, 64), (9, 8, 64)):
        sched = epoch_schedule(n, w, b)
        used = []
        for rows, off, glob in sched:
            assert glob == w * rows and rows >= 2
            for r in range(w):
                used += list(range(off + r * rows, off + (r + 1) * rows))
        assert used == list(range(len(used))) and len(used) <= n and n - len(used) < max(w, 2 * w if not sched else w)
    assert epoch_schedule(9, 8, 64) == []          # one row per rank: training-mode BatchNorm needs two
