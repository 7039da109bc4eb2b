"""End-to-end drop-in test of ``rankaae_amd.Trainer`` on the GPU: same entry point, same files,
same per-epoch metrics as the reference's ``Trainer.from_data(...).train()`` (trainer.py:65-315).
The trajectory is chaotic, so the numbers are compared with the reference's golden run only
statistically (same order of magnitude after the same number of epochs)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from rankaae_amd.synthetic import make_spectra, write_csv


@pytest.mark.parametrize("case,rng_mode", [("fc_small", "host"), ("compact_small", "philox")])
def test_trainer_end_to_end(case, rng_mode, tmp_path):
    from rankaae_amd.parameter import Parameters
    from rankaae_amd.trainer import Trainer
    with open(os.path.join(os.path.dirname(__file__), "golden", f"ref_{case}.json")) as f:
        g = json.load(f)
    cfg = dict(g["config"])
    cfg.update(rng_mode=rng_mode, seed=5, max_epoch=2)
    spec, aux, grid = make_spectra(g["n_rows"], g["n_points"], cfg["n_aux"], seed=g["data_seed"])
    csv = tmp_path / "data.csv"
    write_csv(str(csv), spec, aux, grid)
    lines = []

    class LossLog:
        def info(self, msg):
            lines.append(msg)
    torch.manual_seed(g["model_seed"])
    tr = Trainer.from_data(str(csv), igpu=0, verbose=False, work_dir=str(tmp_path),
                           config_parameters=Parameters(cfg), loss_logger=LossLog())
    seen = []
    metrics = tr.train(callback=lambda ep, m: seen.append((ep, list(m))))
    assert [e for e, _ in seen] == [0, 1] and len(metrics) == 5 and all(np.isfinite(metrics))
    # losses.csv: header + one row for epoch 0 in the reference's format (13 tab-separated fields + trailing ",\t")
    assert lines[0].startswith("Epoch,Train_D,Val_D,Train_G,Val_G,Train_Aux,Val_Aux,Train_Recon")
    row = lines[1]
    assert row.startswith("0,\t") and row.endswith(",\t") and len(row.split(",\t")) == 14
    ref_row = g["losses_csv"][1].split(",\t")
    got_row = row.split(",\t")
    assert got_row[3] == ref_row[3] == "0.000000"          # Train_G column is the constant 0 under gradient reversal
    # same order of magnitude as the reference after 2 epochs: validation reconstruction MSE and rank loss.  With the
    # device RNG the noise / dropout stream is not the reference's, and after only 16 steps the BatchNorm running
    # statistics the validation uses have not settled: over eight seeds the validation MSE of this case spreads from
    # 0.044 to 0.35 around the reference's 0.041 (0.006 ... 0.012 after four epochs, measured in round 3) -- so the
    # device-RNG variant holds the MEDIAN of three seeds to the band (the 8-seed P3 test is the real statistical check)
    ref_m = g["epoch_metrics"][-1]
    recon = [metrics[1]]
    if rng_mode == "philox":
        for extra_seed in (1, 4):
            c2 = dict(cfg, seed=extra_seed)
            torch.manual_seed(g["model_seed"])
            wd2 = tmp_path / f"seed{extra_seed}"
            wd2.mkdir()
            t2 = Trainer.from_data(str(csv), igpu=0, verbose=False, work_dir=str(wd2), config_parameters=Parameters(c2),
                                   loss_logger=type("Q", (), {"info": lambda self, m: None})())
            recon.append(t2.train()[1])
    assert 0.2 * ref_m[1] < float(np.median(recon)) < 5 * ref_m[1], (recon, metrics, ref_m)
    assert 0 < metrics[0] <= 1 and 0 <= metrics[3] <= 1
    # files: final.pt holds whole-module pickles under the reference's keys
    model = torch.load(os.path.join(str(tmp_path), "final.pt"), map_location="cpu", weights_only=False)
    assert set(model) == {"Encoder", "Decoder", "Style Discriminator"}
    enc, dec = model["Encoder"].eval(), model["Decoder"].eval()
    assert dec.nstyle == cfg["nstyle"]
    n_train = int(g["n_rows"] * 0.7)
    x = torch.tensor(spec[n_train:n_train + 16], dtype=torch.float32)
    with torch.no_grad():
        z = enc(x)
        y = dec(z)
    assert z.shape == (16, cfg["nstyle"]) and y.shape == (16, cfg["dim_out"]) and torch.isfinite(y).all()
    # BatchNorm bookkeeping: num_batches_tracked = train-mode forwards (6 per step for the encoder)
    steps = 2 * g["steps_per_epoch"]
    nbt = [v for k, v in enc.state_dict().items() if k.endswith("num_batches_tracked")]
    assert all(int(v) == 6 * steps for v in nbt)
    # the exported (plain PyTorch) encoder reproduces the HIP eval-mode forward used for validation
    val_spec = torch.tensor(spec[n_train:n_train + 105], dtype=torch.float32)
    z_hip, _ = tr.engine.validate(val_spec.to(tr.device), torch.tensor(aux[n_train:n_train + 105],
                                                                       dtype=torch.float32).to(tr.device))
    with torch.no_grad():
        z_ref = enc(val_spec)
    assert torch.allclose(z_hip.cpu(), z_ref, rtol=1e-3, atol=1e-4)
    # the model-selection metrics formed on the device are scipy's on the same styles (trainer.py:286-292)
    import itertools
    from scipy.stats import shapiro, spearmanr
    W, rho = tr.engine.val_style_metrics()
    zc = z_hip.cpu().numpy().T
    assert np.allclose(W, [shapiro(x).statistic for x in zc], rtol=0, atol=1e-12)
    assert np.allclose(rho, [spearmanr(zc[a], zc[b]).correlation
                             for a, b in itertools.combinations(range(len(zc)), 2)], rtol=0, atol=1e-12)


@pytest.mark.parametrize("case,precision", [("p3_fc", "fp32"), ("p3_compact", "fp32"), ("p3_fc", "bf16")])
def test_p3_statistical_parity(case, precision, tmp_path):
    """SURVEY 8d protocol P3: the trajectory is chaotic, so beyond teacher-forced steps the comparison is
    statistical -- 8 model seeds x 6 epochs; the distribution of the final validation reconstruction MSE,
    mean training mutual-information loss and validation rank loss of this engine (device Philox noise) must
    overlap the real reference's (fixtures: oracle/gen_golden.py p3_*): difference of means within
    3 standard errors + 10 %.  ``precision: bf16`` (BASELINE configs[4]'s mixed precision: bf16 STORAGE of the dense
    networks' hidden activations and dropout flags, fp32 arithmetic) is held to the SAME reference distribution and
    the same bound: its per-step gradients are 2-25 % from the fp32 engine's (test_bf16_storage_mode_fc_512_aux12), and
    this is the test that says whether that matters for training."""
    import logging
    from rankaae_amd.parameter import Parameters
    from rankaae_amd.trainer import Trainer
    with open(os.path.join(os.path.dirname(__file__), "golden", f"ref_{case}.json")) as f:
        g = json.load(f)
    ref = np.array([r["final_metrics"] for r in g["runs"]])
    spec, aux, _ = make_spectra(g["n_rows"], g["n_points"], g["config"]["n_aux"], seed=g["data_seed"])
    quiet = logging.getLogger("p3_quiet")
    quiet.addHandler(logging.NullHandler())
    quiet.propagate = False
    got = []
    for r in g["runs"]:
        cfg = dict(g["config"])
        cfg.update(rng_mode="philox", seed=r["model_seed"], precision=precision)
        torch.manual_seed(r["model_seed"])
        wd = tmp_path / f"s{r['model_seed']}"
        wd.mkdir()
        tr = Trainer.from_data(None, igpu=0, verbose=False, work_dir=str(wd), config_parameters=Parameters(cfg),
                               logger=quiet, loss_logger=quiet, arrays=(spec, aux))
        got.append([float(x) for x in tr.train()])
    got = np.array(got)
    assert np.all(np.isfinite(got))
    n = len(ref)
    for j, name in ((1, "val recon MSE"), (2, "mean train MI"), (4, "val rank loss")):
        se = np.sqrt((ref[:, j].var(ddof=1) + got[:, j].var(ddof=1)) / n)
        diff = abs(got[:, j].mean() - ref[:, j].mean())
        assert diff <= 3 * se + 0.1 * abs(ref[:, j].mean()), \
            f"{case} {precision} {name}: ours {got[:, j].mean():.5f}+-{got[:, j].std(ddof=1):.5f} vs reference " \
            f"{ref[:, j].mean():.5f}+-{ref[:, j].std(ddof=1):.5f}"
        print(f"\n{case} {precision} {name}: ours {got[:, j].mean():.5f}+-{got[:, j].std(ddof=1):.5f} vs reference "
              f"{ref[:, j].mean():.5f}+-{ref[:, j].std(ddof=1):.5f} (allowed difference of means {3 * se + 0.1 * abs(ref[:, j].mean()):.5f})")
    # Shapiro W and the coupling metric live in [0, 1]; same ballpark
    assert abs(got[:, 0].mean() - ref[:, 0].mean()) < 0.15 and abs(got[:, 3].mean() - ref[:, 3].mean()) < 0.25


def test_latent_export_matches_exported_modules(tmp_path):
    """``Reconstruct`` (sc/report/analysis_new.py:94-129) on the HIP engine: the three text files in
    ``np.savetxt``'s default format, one row per spectrum, equal to what the saved ``final.pt`` modules (plain
    PyTorch forward, the reference report's own consumer) compute."""
    import logging
    from rankaae_amd.export import Reconstruct
    from rankaae_amd.parameter import Parameters
    from rankaae_amd.trainer import Trainer
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_compact_small.json")) as f:
        g = json.load(f)
    cfg = dict(g["config"])
    cfg.update(rng_mode="philox", seed=3, max_epoch=1)
    spec, aux, _ = make_spectra(g["n_rows"], g["n_points"], cfg["n_aux"], seed=g["data_seed"])
    quiet = logging.getLogger("export_quiet")
    quiet.addHandler(logging.NullHandler())
    quiet.propagate = False
    torch.manual_seed(7)
    tr = Trainer.from_data(None, igpu=0, verbose=False, work_dir=str(tmp_path), config_parameters=Parameters(cfg),
                           logger=quiet, loss_logger=quiet, arrays=(spec, aux))
    tr.train()

    class DS:
        pass
    ds = DS()
    ds.spec = spec[-105:]                       # the test split of the reference's 70/15/15 cut
    ev = Reconstruct(name="recon")
    res = ev.evaluate(ds, tr, path_to_save=str(tmp_path))
    styles = np.loadtxt(tmp_path / "recon_styles.txt")
    spec_out = np.loadtxt(tmp_path / "recon_spec_out.txt")
    spec_in = np.loadtxt(tmp_path / "recon_spec_in.txt")
    assert styles.shape == (105, cfg["nstyle"]) and spec_out.shape == (105, cfg["dim_out"])
    assert np.array_equal(spec_in.astype(np.float32), ds.spec.astype(np.float32))
    with open(tmp_path / "recon_styles.txt") as f:
        first = f.readline().split(" ")
    assert len(first) == cfg["nstyle"] and len(first[0].strip()) in (24, 25) and "e" in first[0]   # %.18e
    model = torch.load(os.path.join(str(tmp_path), "final.pt"), map_location="cpu", weights_only=False)
    enc, dec = model["Encoder"].eval(), model["Decoder"].eval()
    with torch.no_grad():
        z = enc(torch.tensor(ds.spec, dtype=torch.float32))
        y = dec(z)
    assert np.allclose(styles, z.numpy(), rtol=1e-3, atol=1e-4)
    assert np.allclose(spec_out, y.numpy(), rtol=1e-3, atol=1e-4)
    assert np.array_equal(res["styles"], styles.astype(np.float32))


def test_decoder_sweeps_match_exported_decoder(tmp_path):
    """``spectra_variation`` (the numbers of the report's ``plot_spectra_variation``, sc/report/analysis.py:33-86)
    on the HIP engine against the saved plain-PyTorch decoder: exact construction for ``n_sampling == 0``; for
    ``n_sampling > 0`` the same draws are decoded by both (the generator is re-seeded)."""
    import logging
    from rankaae_amd.export import spectra_variation
    from rankaae_amd.parameter import Parameters
    from rankaae_amd.trainer import Trainer
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_compact_small.json")) as f:
        g = json.load(f)
    cfg = dict(g["config"])
    cfg.update(rng_mode="philox", seed=3, max_epoch=1)
    spec, aux, _ = make_spectra(g["n_rows"], g["n_points"], cfg["n_aux"], seed=g["data_seed"])
    quiet = logging.getLogger("sweep_quiet")
    quiet.addHandler(logging.NullHandler())
    quiet.propagate = False
    torch.manual_seed(7)
    tr = Trainer.from_data(None, igpu=0, verbose=False, work_dir=str(tmp_path), config_parameters=Parameters(cfg),
                           logger=quiet, loss_logger=quiet, arrays=(spec, aux))
    tr.train()
    dec = torch.load(os.path.join(str(tmp_path), "final.pt"), map_location="cpu", weights_only=False)["Decoder"].eval()
    ns = cfg["nstyle"]
    styles = np.random.default_rng(0).standard_normal((300, ns))
    left, right = np.percentile(styles[:, 1], [5, 95])
    c, out = spectra_variation(tr, 1, styles, n_spec=20, n_sampling=0)
    assert np.allclose(c, np.linspace(left, right, 20)) and out.shape == (20, cfg["dim_out"])
    con = torch.zeros(20, ns)
    con[:, 1] = torch.tensor(c, dtype=torch.float)
    with torch.no_grad():
        ref = dec(con).reshape(20, -1).numpy()
    assert np.allclose(out, ref, rtol=1e-3, atol=1e-4)
    torch.cuda.manual_seed(11)
    v, out = spectra_variation(tr, 2, styles, n_spec=7, n_sampling=33)
    torch.cuda.manual_seed(11)
    con = torch.randn([7, 33, ns], device=tr.device)
    con[..., 2] = torch.linspace(*np.percentile(styles[:, 2], [5, 95]), 7, device=tr.device)[:, None]
    with torch.no_grad():
        ref = dec(con.reshape(7 * 33, ns).cpu()).reshape(7, 33, -1).mean(axis=1).numpy()
    assert out.shape == (7, cfg["dim_out"]) and np.allclose(out, ref, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("workers", [1, 2])
def test_train_sc_command_line_and_trial_workers(workers, tmp_path):
    """``train_sc -c cfg.yaml -w dir`` (sc/cmd/train_sc.py:105-157): two trials, sequential and through the
    per-GPU worker processes (both workers share this box's one GPU), leave the reference's directory layout."""
    import subprocess
    import sys
    import yaml
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_compact_small.json")) as f:
        g = json.load(f)
    cfg = dict(g["config"])
    cfg.update(max_epoch=2, trials=2, data_file="data.csv", verbose=False, timeout=1, trial_mode="processes")
    spec, aux, grid = make_spectra(g["n_rows"], g["n_points"], cfg["n_aux"], seed=g["data_seed"])
    write_csv(str(tmp_path / "data.csv"), spec, aux, grid)
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    env = dict(os.environ, RANKAAE_TRIAL_WORKERS=str(workers),
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "rankaae_amd.cmd.train_sc", "-c", "cfg.yaml", "-w", str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    main_log = (tmp_path / "main_process_message.txt").read_text()
    assert "START" in main_log and "END" in main_log and f"Running with {workers} process(es)." in main_log
    assert "for 2 trails" in main_log
    for k in (1, 2):
        job = tmp_path / "training" / f"job_{k}"
        # (best.pt appears only when the combined metric beats the reference's initial guess of 10.0)
        for name in ("messages.txt", "losses.csv", "final.pt", "checkpoints"):
            assert (job / name).exists(), (k, name)
        assert (job / "losses.csv").read_text().startswith("Epoch,Train_D,Val_D")
        assert "Training finished" in (job / "messages.txt").read_text()
        model = torch.load(job / "final.pt", map_location="cpu", weights_only=False)
        assert set(model) == {"Encoder", "Decoder", "Style Discriminator"}


@pytest.mark.parametrize("case,mode", [("compact_small", "threads"), ("fc_small", "threads"), ("fc_small", "batched"),
                                       ("fc_small", "auto"), ("compact_small", "batched"), ("compact_small", "auto")])
def test_train_sc_concurrent_trials_equal_the_trial_alone(case, mode, tmp_path):
    """VERDICT r2 item 3 (the reference's real workload: ``trials: 8`` in example/fix_config.yaml): the
    trials of a run share the GPU -- ``trial_mode: threads`` runs them in threads of one process, each with its own
    engine, HIP stream, captured graph and host generator (seed ``trial_seed + k``); ``trial_mode: batched`` (what
    ``auto`` picks) trains them in lockstep, every step of the group ONE launch sequence with
    ``gridDim.z = trials``.  What runs beside a trial must not change it: trial 2 of a three-trial concurrent run ends
    with BITWISE the weights (and metrics) of the same trial run alone, and the reference's directory layout is
    written for every trial."""
    import subprocess
    import sys
    import yaml
    with open(os.path.join(os.path.dirname(__file__), "golden", f"ref_{case}.json")) as f:
        g = json.load(f)
    spec, aux, grid = make_spectra(g["n_rows"], g["n_points"], g["config"]["n_aux"], seed=g["data_seed"])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RANKAAE_TRIAL_WORKERS", "RANKAAE_TRIALS_PER_GPU"):
        env.pop(k, None)
    models = {}
    for name, over in (("together", dict(trials=3, trials_per_gpu=3, trial_seed=700)), ("alone", dict(trials=1, trial_seed=701))):
        wd = tmp_path / name
        wd.mkdir()
        cfg = dict(g["config"])
        cfg.update(max_epoch=3, data_file="data.csv", verbose=False, timeout=1, trial_mode=mode, **over)
        write_csv(str(wd / "data.csv"), spec, aux, grid)
        with open(wd / "cfg.yaml", "w") as f:
            yaml.safe_dump(cfg, f)
        r = subprocess.run([sys.executable, "-m", "rankaae_amd.cmd.train_sc", "-c", "cfg.yaml", "-w", str(wd)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        main_log = (wd / "main_process_message.txt").read_text()
        assert f"Running with {over['trials']} process(es)." in main_log, main_log
        for k in range(1, over["trials"] + 1):
            job = wd / "training" / f"job_{k}"
            assert "Training finished" in (job / "messages.txt").read_text()
            models[(name, k)] = torch.load(job / "final.pt", map_location="cpu", weights_only=False)
    a, b = models[("together", 2)], models[("alone", 1)]
    for key in ("Encoder", "Decoder", "Style Discriminator"):
        sa, sb = a[key].state_dict(), b[key].state_dict()
        assert sa.keys() == sb.keys()
        for name in sa:
            assert torch.equal(sa[name], sb[name]), (key, name)
    # ... and the trials of one run differ from each other (different seeds)
    c = models[("together", 1)]["Encoder"].state_dict()
    assert any(not torch.equal(c[n], a["Encoder"].state_dict()[n]) for n in c)


def test_train_sc_auto_mode_falls_back_to_threads_when_batching_is_refused(tmp_path):
    """``trial_mode: auto`` picks the batched launch sequence; a configuration whose step meets a kernel without the
    batched form (here: ``fused_blocks: false``, the per-layer conv kernels) is refused by the recorder at the first
    step -- ``auto`` then trains the group in threads, from the start, and says so; ``batched`` raises."""
    import subprocess
    import sys
    import yaml
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_compact_small.json")) as f:
        g = json.load(f)
    spec, aux, grid = make_spectra(g["n_rows"], g["n_points"], g["config"]["n_aux"], seed=g["data_seed"])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RANKAAE_TRIAL_WORKERS", "RANKAAE_TRIALS_PER_GPU"):
        env.pop(k, None)
    for mode, ok in (("auto", True), ("batched", False)):
        wd = tmp_path / mode
        wd.mkdir()
        cfg = dict(g["config"])
        cfg.update(max_epoch=2, data_file="data.csv", verbose=False, timeout=1, trial_mode=mode, trials=2, trial_seed=5,
                   fused_blocks=False)
        write_csv(str(wd / "data.csv"), spec, aux, grid)
        with open(wd / "cfg.yaml", "w") as f:
            yaml.safe_dump(cfg, f)
        r = subprocess.run([sys.executable, "-m", "rankaae_amd.cmd.train_sc", "-c", "cfg.yaml", "-w", str(wd)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert (r.returncode == 0) == ok, r.stderr[-3000:]
        if ok:
            assert "batched launches refused" in (wd / "main_process_message.txt").read_text() + r.stderr
            for k in (1, 2):
                assert "Training finished" in (wd / "training" / f"job_{k}" / "messages.txt").read_text()
                assert (wd / "training" / f"job_{k}" / "final.pt").exists()
