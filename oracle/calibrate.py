#!/usr/bin/env python
"""Calibration of the CPU baseline: the oracle ("port") next to the REAL reference.

TEST INFRASTRUCTURE -- runs only in the build container, where ``/root/reference`` exists.  ``bench.py`` times the
oracle on the GPU box's host cores (the reference cannot travel); this script times both on the SAME machine, same
synthetic 7000x256 spectra, batch 256, 1 thread, autograd anomaly detection as shipped (on), train-only steps/s, and
writes the ratio to ``profiles/cpu_calibration.json`` so that the port's number can be read as the reference's.

Usage:  python oracle/calibrate.py [seconds per leg, default 20]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import gen_golden, ref_train                      # noqa: E402
from rankaae_amd.synthetic import make_spectra, write_csv     # noqa: E402


def time_reference(cfg, spec, aux, grid, budget):
    """``Trainer.train``'s batch loop of the unmodified reference (trainer.py:103-204), driven batch by batch."""
    import logging
    from sc.clustering import trainer as T
    from sc.utils.parameter import Parameters
    orig_sched = T.ReduceLROnPlateau
    T.ReduceLROnPlateau = lambda opt, **kw: orig_sched(opt, **{k: v for k, v in kw.items() if k != "verbose"})
    tmp = tempfile.mkdtemp(prefix="raae_calib_")
    csv = os.path.join(tmp, "data.csv")
    write_csv(csv, spec, aux, grid)
    log = logging.getLogger("calib")
    log.addHandler(logging.NullHandler())
    log.propagate = False
    torch.manual_seed(1234)
    c = dict(cfg, max_epoch=10 ** 6)
    tr = T.Trainer.from_data(csv, igpu=0, verbose=False, work_dir=tmp, config_parameters=Parameters(c), logger=log,
                             loss_logger=log)
    steps = {"n": 0, "t0": None}

    class Stop(Exception):
        pass
    orig = T.adversarial_loss            # called once per training step (phase A) and once per validation

    def counted(*a, **kw):
        if steps["t0"] is None:
            steps["t0"] = time.perf_counter()
        elif time.perf_counter() - steps["t0"] > budget and steps["n"] >= 3:
            raise Stop
        steps["n"] += 1
        return orig(*a, **kw)
    T.adversarial_loss = counted
    try:
        tr.train()
    except Stop:
        pass
    finally:
        T.adversarial_loss, T.ReduceLROnPlateau = orig, orig_sched
    el = time.perf_counter() - steps["t0"]
    return (steps["n"] - 1) / el, steps["n"] - 1, el      # the first call only starts the clock


def time_oracle(cfg, spec, aux, budget, anomaly=True):
    torch.manual_seed(1234)
    tr = ref_train.OracleTrainer(spec, aux, cfg)
    for m in (tr.encoder, tr.decoder, tr.discriminator):
        m.train()
    perm = ref_train.epoch_permutation(len(tr.train_spec)).numpy()
    bs = cfg["batch_size"]
    prev = torch.is_anomaly_enabled()
    torch.autograd.set_detect_anomaly(anomaly)
    t0, n = time.perf_counter(), 0
    while True:
        rows = perm[(n % 19) * bs:(n % 19 + 1) * bs]
        tr.train_step(torch.tensor(tr.train_spec[rows], dtype=torch.float32),
                      torch.tensor(tr.train_aux[rows], dtype=torch.float32), 0.3, 0)
        n += 1
        el = time.perf_counter() - t0
        if el > budget and n >= 3:
            break
    torch.autograd.set_detect_anomaly(prev)
    return n / el, n, el


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    gen_golden._install_shims()
    import sc.clustering.trainer  # noqa: F401   (sets anomaly detection on at import, trainer.py:11)
    torch.set_num_threads(1)
    out = {"cpu_model": gen_golden._cpu_model(), "threads": 1, "torch": torch.__version__, "rows": 7000, "batch": 256,
           "seconds_per_leg": budget, "note": "train-only steps/s, anomaly detection on (as the reference ships)"}
    spec, aux, grid = make_spectra(7000, 256, 5, seed=0)
    for ae in ("compact", "FC"):
        cfg = dict(gen_golden.BASE_CONFIG, ae_form=ae, batch_size=256)
        ref = time_reference(cfg, spec, aux, grid, budget)
        port = time_oracle(cfg, spec, aux, budget, anomaly=True)
        port_off = time_oracle(cfg, spec, aux, budget / 2, anomaly=False)
        out[ae] = {"reference_steps_per_s": round(ref[0], 3), "port_steps_per_s": round(port[0], 3),
                   "port_over_reference": round(port[0] / ref[0], 4), "port_anomaly_off_steps_per_s": round(port_off[0], 3),
                   "reference_steps": ref[1], "port_steps": port[1]}
        print(ae, out[ae], flush=True)
    with open(os.path.join(REPO, "profiles", "cpu_calibration.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
