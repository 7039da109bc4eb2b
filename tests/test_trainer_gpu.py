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
    # same order of magnitude as the reference after 2 epochs: validation reconstruction MSE and rank loss
    ref_m = g["epoch_metrics"][-1]
    assert 0.2 * ref_m[1] < metrics[1] < 5 * ref_m[1], (metrics, ref_m)
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
