#!/usr/bin/env python
"""Generate golden vectors by running the REAL reference (``/root/reference``) on CPU.

TEST INFRASTRUCTURE -- not product code.  Runs only in the build container (the
reference cannot travel to the GPU box).  It imports ``sc.clustering.trainer``
unmodified, with three empty stand-in modules for packages that are absent here and
are not on the hot path (``seaborn``: plotting; ``torch_optimizer``: AdaBound/RAdam
registry entries; ``torchvision.transforms.Compose``: a 3-line list-apply), plus a
wrapper dropping the ``verbose=`` kwarg that torch 2.10's ``ReduceLROnPlateau``
no longer accepts.  Loss functions are wrapped with *recorders*
(the wrapped function is the reference's own) so every per-step value is captured.

Output: ``tests/golden/ref_<name>.json`` -- inputs are reproducible from
``rankaae_amd.synthetic.make_spectra`` + the config stored in the fixture; expected
outputs are the reference's per-step phase losses, per-epoch validation losses and
metrics, parameter checksums at init and at the end, and a slice of validation styles.

Usage:  python oracle/gen_golden.py            (writes all fixtures)
"""
import json
import logging
import os
import platform
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REFERENCE = os.environ.get("RANKAAE_REFERENCE", "/root/reference")


def _install_shims():
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    to = types.ModuleType("torch_optimizer")
    to.AdaBound = object
    to.RAdam = object
    sys.modules.setdefault("torch_optimizer", to)
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x
    tr.Compose = Compose
    tv.transforms = tr
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tr)
    sys.path.insert(0, REFERENCE)


BASE_CONFIG = dict(  # example/fix_config.yaml of the reference, verbatim values
    trials=1, timeout=10, verbose=False, max_epoch=2, batch_size=64,
    gradient_reversal=True, alpha_flat_step=739, alpha_limit=0.7172,
    decoder_activation="Softplus", dis_beta=1.1, dis_dropout_rate=0.056, dis_noise=0.56,
    gen_beta=1.1, n_aux=5, nstyle=6, ae_form="FC", dim_in=256, dim_out=256, n_layers=5,
    FC_discriminator_layers=3, use_cnn_discriminator=False, dropout_rate=0.04,
    sch_factor=0.1, sch_patience=100, lr_base=0.001, lr_ratio_Corr=10, lr_ratio_Mutual=1,
    lr_ratio_Reconn=10, lr_ratio_Smooth=1, lr_ratio_dis=1, lr_ratio_gen=10,
    optimizer_name="AdamW", spec_noise=0.02, use_flex_spec_target=True, weight_decay=0.01,
    kendall_activation=True, epoch_stop_smooth=1500,
)

CASES = {
    # name: (n_rows, n_points, overrides, data_seed, model_seed)
    "fc_small": (700, 256, dict(ae_form="FC", batch_size=64, max_epoch=2), 0, 1234),
    "compact_small": (700, 256, dict(ae_form="compact", batch_size=64, max_epoch=2), 0, 1234),
    "fc_c2": (7000, 256, dict(ae_form="FC", batch_size=256, max_epoch=1), 0, 1234),
    "compact_c2": (7000, 256, dict(ae_form="compact", batch_size=256, max_epoch=1), 0, 1234),
    "fc_adam_nodrop": (700, 256, dict(ae_form="FC", batch_size=128, max_epoch=1, optimizer_name="Adam",
                                      dropout_rate=0.0, dis_dropout_rate=0.0, kendall_activation=False,
                                      use_flex_spec_target=False, epoch_stop_smooth=0), 3, 7),
    "fc_512_aux12": (350, 512, dict(ae_form="FC", batch_size=64, max_epoch=1, dim_in=512, dim_out=512,
                                    n_aux=12, nstyle=13), 1, 99),
    # lr_base = 0: Adam/AdamW leave every weight where it is, so the trajectory is NOT chaotic (SURVEY finding 8
    # is about the updates) while everything else still runs -- the five forward/backward phases of every step incl.
    # the ragged last batch, the random tape, the BatchNorm running statistics of 6 + 4 train-mode forwards per
    # step, and the per-epoch validation pass in eval mode on those running statistics.  A FREE-RUNNING
    # implementation can be held to these values over two whole epochs (tests/test_engine_gpu.py::test_p4_*).
    "fc_frozen": (700, 256, dict(ae_form="FC", batch_size=64, max_epoch=2, lr_base=0.0), 5, 4321),
    "compact_frozen": (700, 256, dict(ae_form="compact", batch_size=64, max_epoch=2, lr_base=0.0), 5, 4321),
    # BASELINE configs[0]: example/fix_config.yaml as it stands (FC, batch 1024, 7000 x 256), one epoch free and two
    # epochs frozen (five steps per epoch, the last one the ragged 804-row batch).
    "fc_example": (7000, 256, dict(ae_form="FC", batch_size=1024, max_epoch=1), 2, 2022),
    "fc_example_frozen": (7000, 256, dict(ae_form="FC", batch_size=1024, max_epoch=2, lr_base=0.0), 2, 2022),
}


def checksum(module):
    out = {}
    for k, v in module.state_dict().items():
        v = v.detach().double()
        out[k] = [float(v.sum()), float(v.abs().sum())]
    return out


def run_case(name, case=None, write=True):
    from sc.clustering import trainer as T
    from sc.utils.parameter import Parameters
    from rankaae_amd.synthetic import make_spectra, write_csv

    n_rows, n_points, over, data_seed, model_seed = case or CASES[name]
    cfg = dict(BASE_CONFIG)
    cfg.update(over)
    spec, aux, grid = make_spectra(n_rows, n_points, cfg["n_aux"], seed=data_seed)
    tmp = tempfile.mkdtemp(prefix="raae_golden_")
    csv = os.path.join(tmp, "data.csv")
    write_csv(csv, spec, aux, grid)

    rec = {k: [] for k in ("adversarial", "kendall", "recon", "mutual_info", "smooth")}
    names = {"adversarial_loss": "adversarial", "kendall_constraint": "kendall", "recon_loss": "recon",
             "mutual_info_loss": "mutual_info", "smoothness_loss": "smooth"}
    originals = {}
    for fn, key in names.items():
        orig = getattr(T, fn)
        originals[fn] = orig

        def make(orig=orig, key=key):
            def wrapper(*a, **kw):
                out = orig(*a, **kw)
                rec[key].append(float(out.detach()))
                return out
            return wrapper
        setattr(T, fn, make())
    orig_sched = T.ReduceLROnPlateau

    def sched(opt, **kw):
        kw.pop("verbose", None)
        return orig_sched(opt, **kw)
    T.ReduceLROnPlateau = sched

    torch.set_num_threads(1)
    torch.manual_seed(model_seed)
    log = logging.getLogger("golden_" + name)
    log.addHandler(logging.NullHandler())
    log.propagate = False
    lines = []

    class LossLog:
        def info(self, msg):
            lines.append(msg)
    trainer = T.Trainer.from_data(csv, igpu=0, verbose=False, work_dir=tmp,
                                  config_parameters=Parameters(dict(cfg)), logger=log, loss_logger=LossLog())
    init = {"Encoder": checksum(trainer.encoder), "Decoder": checksum(trainer.decoder),
            "Style Discriminator": checksum(trainer.discriminator)}
    epoch_metrics = []
    metrics = trainer.train(callback=lambda ep, m: epoch_metrics.append([float(x) for x in m]))
    final = {"Encoder": checksum(trainer.encoder), "Decoder": checksum(trainer.decoder),
             "Style Discriminator": checksum(trainer.discriminator)}

    # eval-mode styles of the first 8 validation rows with the final encoder
    trainer.encoder.eval()
    n_train = int(n_rows * 0.7)
    val_spec = torch.tensor(spec[n_train:n_train + 8], dtype=torch.float32)
    with torch.no_grad():
        styles = trainer.encoder(val_spec).double().numpy().tolist()

    for fn, orig in originals.items():
        setattr(T, fn, orig)
    T.ReduceLROnPlateau = orig_sched
    bn_buffers = {nm: {k: v.double().numpy().tolist() for k, v in mod.state_dict().items()
                       if k.endswith("running_mean") or k.endswith("running_var")}
                  for nm, mod in (("Encoder", trainer.encoder), ("Decoder", trainer.decoder))}

    n_train_rows = int(n_rows * 0.7)
    steps_per_epoch = -(-n_train_rows // cfg["batch_size"])
    fixture = {
        "case": name, "config": cfg, "n_rows": n_rows, "n_points": n_points, "data_seed": data_seed,
        "model_seed": model_seed, "steps_per_epoch": steps_per_epoch,
        "torch": torch.__version__, "cpu": platform.processor() or platform.machine(),
        "cpu_model": _cpu_model(), "threads": 1,
        # per-call values in call order; train and validation calls interleave per epoch:
        # each epoch = steps_per_epoch training calls then exactly one validation call.
        "loss_calls": rec, "epoch_metrics": epoch_metrics, "final_metrics": [float(x) for x in metrics],
        "losses_csv": lines, "init_checksum": init, "final_checksum": final, "val_styles_first8": styles,
    }
    if name.endswith("_frozen"):
        fixture["final_bn_buffers"] = bn_buffers
    if not write:
        return fixture
    out = os.path.join(REPO, "tests", "golden", f"ref_{name}.json")
    with open(out, "w") as f:
        json.dump(fixture, f, indent=1)
    print("wrote", out, {k: v[:2] for k, v in rec.items()})


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


P3_SEEDS = [11, 22, 33, 44, 55, 66, 77, 88]
P3_CASES = {
    # SURVEY 8d protocol P3: >= 8 model seeds x fixed epochs; the DISTRIBUTION of the final metrics is the target
    "p3_fc": (700, 256, dict(ae_form="FC", batch_size=64, max_epoch=6), 0),
    "p3_compact": (700, 256, dict(ae_form="compact", batch_size=64, max_epoch=6), 0),
}


def run_p3(name):
    n_rows, n_points, over, data_seed = P3_CASES[name]
    runs = []
    for seed in P3_SEEDS:
        fx = run_case(f"{name}_{seed}", case=(n_rows, n_points, over, data_seed, seed), write=False)
        runs.append({"model_seed": seed, "final_metrics": fx["final_metrics"], "epoch_metrics": fx["epoch_metrics"]})
        print(name, seed, fx["final_metrics"])
    cfg = dict(BASE_CONFIG)
    cfg.update(over)
    out = os.path.join(REPO, "tests", "golden", f"ref_{name}.json")
    with open(out, "w") as f:
        json.dump({"case": name, "config": cfg, "n_rows": n_rows, "n_points": n_points, "data_seed": data_seed,
                   "metric_names": ["min_shapiro_W", "val_recon_mse", "mean_train_mi", "max_abs_spearman", "val_rank_loss"],
                   "runs": runs, "torch": torch.__version__, "cpu_model": _cpu_model(), "threads": 1}, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    _install_shims()
    which = sys.argv[1:] or list(CASES)
    for nm in which:
        if nm in P3_CASES:
            run_p3(nm)
        else:
            run_case(nm)
