#!/usr/bin/env python
"""Benchmark of the RankAAE training-step hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--ae-form compact|FC] [--batch 256]

With ``--gpus N`` and no launcher (WORLD_SIZE unset) it starts its N rank processes itself.
One "step" = the reference's five-phase update (adversarial, rank, reconstruction,
mutual-information, smoothness: 6 encoder + 4 decoder forwards, 5 backwards, 5 AdamW
updates; ``sc/clustering/trainer.py:103-204``) on one batch of 256 synthetic 256-point
spectra drawn from a device-resident 7000-row dataset (BASELINE.json ``configs[1]``).
Rank 0 prints ONE JSON line.  ``value`` = batch-256 steps per second summed over all
ranks (weak scaling: every GPU steps its own 256-row shard of a 256*N global batch and
the five gradient arenas are averaged over RCCL).  The line also carries:
  roofline      the dominant kernel of this workload: algorithmic bytes / HIP-event time
  cpu_baseline  the CPU oracle (PyTorch fp32 autograd restatement of the reference) timed on
                this box's host cores: 1 thread + anomaly detection on, as the reference ships
"""
import argparse
import gc
import json
import os
import sys
import time


import numpy as np

os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))     # before the HIP runtime initialises: rankaae_amd/__init__.py says why
import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BASE_CFG = dict(  # example/fix_config.yaml of the reference; only trials/batch_size/ae_form/max_epoch change
    trials=1, timeout=10, verbose=False, max_epoch=2000, batch_size=256,
    gradient_reversal=True, alpha_flat_step=739, alpha_limit=0.7172, decoder_activation="Softplus",
    dis_beta=1.1, dis_dropout_rate=0.056, dis_noise=0.56, gen_beta=1.1, n_aux=5, nstyle=6, ae_form="compact",
    dim_in=256, dim_out=256, n_layers=5, FC_discriminator_layers=3, use_cnn_discriminator=False,
    dropout_rate=0.04, sch_factor=0.1, sch_patience=100, lr_base=0.001, lr_ratio_Corr=10, lr_ratio_Mutual=1,
    lr_ratio_Reconn=10, lr_ratio_Smooth=1, lr_ratio_dis=1, lr_ratio_gen=10, optimizer_name="AdamW",
    spec_noise=0.02, use_flex_spec_target=True, weight_decay=0.01, kendall_activation=True, epoch_stop_smooth=1500)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def build_models(cfg, seed):
    from rankaae_amd import model as pm
    torch.manual_seed(seed)
    cls = pm.AE_CLS_DICT[cfg["ae_form"]]
    enc = cls["encoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], dim_in=cfg["dim_in"],
                         n_layers=cfg["n_layers"])
    dec = cls["decoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"],
                         last_layer_activation=cfg["decoder_activation"], dim_out=cfg["dim_out"],
                         n_layers=cfg["n_layers"])
    dis = pm.DiscriminatorFC(nstyle=cfg["nstyle"], dropout_rate=cfg["dis_dropout_rate"], noise=cfg["dis_noise"],
                             layers=cfg["FC_discriminator_layers"])
    return enc, dec, dis


def cpu_baseline(cfg, spec, aux, budget_s):
    """The oracle (kind "port") on this box's host cores, 1 thread, anomaly detection ON
    (how ``train_sc`` runs the reference: train_sc.py:68-70, trainer.py:11); also times it with
    anomaly detection off.  Bounded sample: whole steps until ``budget_s`` seconds are used."""
    from oracle import ref_train
    nthreads = torch.get_num_threads()
    out = {}
    # legs: (i) 1 thread + anomaly detection on = how the reference ships, (ii) 1 thread, anomaly off,
    # (iii) all cores torch gives this process (anomaly off; the reference pins 1 thread only when interop > 2,
    # train_sc.py:68-70)
    for anomaly, share, threads in ((True, 0.5, 1), (False, 0.25, 1), ("all", 0.25, nthreads)):
        torch.set_num_threads(threads)
        torch.manual_seed(1234)
        tr = ref_train.OracleTrainer(spec, aux, cfg)
        for m in (tr.encoder, tr.decoder, tr.discriminator):
            m.train()
        perm = ref_train.epoch_permutation(len(tr.train_spec)).numpy()
        bs = cfg["batch_size"]
        prev = torch.is_anomaly_enabled()
        torch.autograd.set_detect_anomaly(anomaly is True)
        t0, n = time.perf_counter(), 0
        while True:
            rows = perm[(n % 19) * bs:(n % 19 + 1) * bs]
            tr.train_step(torch.tensor(tr.train_spec[rows], dtype=torch.float32),
                          torch.tensor(tr.train_aux[rows], dtype=torch.float32), 0.3, 0)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s * share and n >= 3:
                break
        torch.autograd.set_detect_anomaly(prev)
        out[anomaly] = (n / el, n, el)
    torch.set_num_threads(nthreads)
    v, n, el = out[True]
    return {"value": round(v, 3), "unit": "steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} steps of the same workload in {el:.1f} s, 1 thread, autograd anomaly detection on "
                      f"(as the reference ships); anomaly off: {out[False][0]:.3f} steps/s",
            "anomaly_off": {"value": round(out[False][0], 3), "cores": 1, "steps": out[False][1]},
            "all_cores": {"value": round(out["all"][0], 3), "threads": nthreads, "steps": out["all"][1],
                          "note": "torch intra-op threads = torch.get_num_threads() of this process, anomaly detection off"},
            "host_cpus": os.cpu_count()}


def epoch_inclusive(cfg, spec, aux, epochs=8):
    """SURVEY 8d asks for train-only AND epoch-inclusive throughput: ``Trainer.train`` with its per-epoch
    validation forward, five validation losses, Shapiro / Spearman metrics (raae_style_metrics, on the device) and
    the scheduler step; epochs 3.. are timed (the first two emit and capture the graphs)."""
    import logging
    import tempfile
    from rankaae_amd.parameter import Parameters
    from rankaae_amd.trainer import Trainer
    quiet = logging.getLogger("bench_quiet")
    quiet.addHandler(logging.NullHandler())
    quiet.propagate = False
    c = dict(cfg)
    c.update(max_epoch=epochs, rng_mode="philox", seed=1234)
    stamps = []
    with tempfile.TemporaryDirectory() as wd:
        tr = Trainer.from_data(None, igpu=torch.cuda.current_device(), verbose=False, work_dir=wd,
                               config_parameters=Parameters(c), logger=quiet, loss_logger=quiet, arrays=(spec, aux))
        steps_per_epoch = len(tr.train_loader)
        tr.train(callback=lambda ep, m: (torch.cuda.synchronize(), stamps.append(time.perf_counter())))
    dt = stamps[-1] - stamps[1]                     # epochs 2 .. epochs-1
    n = (len(stamps) - 2) * steps_per_epoch
    return {"value": round(n / dt, 2), "unit": "steps/s including per-epoch validation, metrics and scheduler",
            "steps_per_epoch": steps_per_epoch, "epochs_timed": len(stamps) - 2, "ms_per_epoch": round(1e3 * dt / (len(stamps) - 2), 2)}


def _profile(pattern):
    """Newest committed profile file matching ``profiles/r<round>_<pattern>`` (this round's first)."""
    for rnd in ("r3", "r2"):
        path = os.path.join(REPO, "profiles", f"{rnd}_{pattern}")
        if os.path.exists(path):
            return path
    return None


def pmc_traffic(kernel, ae_form, b):
    """HBM bytes per launch of the kernel FAMILY ``kernel`` (all template instances, launch-weighted) from the committed
    rocprofv3 PMC summary (FETCH_SIZE and WRITE_SIZE collected in separate ``--pmc`` passes of this same command and
    summarised by tools/pmc_summary.py into profiles/r*_pmc_traffic_*.json).  FETCH_SIZE is DOUBLED as
    MI355X_MICROARCH.md prescribes for gfx950's 16-byte-per-lane streaming reads (it tallies 128-byte requests at
    64 B); WRITE_SIZE is exact.  None when there is no summary for this workload or kernel.  Hardware counters cannot
    be read from inside the process: this is the same command's committed profile, not a measurement of this run."""
    path = _profile(f"pmc_traffic_{ae_form.lower()}_b{b}.json")
    if path is None:
        return None
    with open(path) as f:
        kernels = json.load(f)["kernels"]
    fam = kernel.split("<")[0].split("[")[0]
    k = kernels.get(fam)
    if k is None:           # a family with one instance is stored under the instance's name
        hits = [v for name, v in kernels.items() if name.split("<")[0] == fam]
        n = sum(v["launches"] for v in hits)
        if not n:
            return None
        k = {"hbm_bytes_fetch_doubled": sum(v["hbm_bytes_fetch_doubled"] * v["launches"] for v in hits) / n,
             "hbm_bytes_raw": sum(v["hbm_bytes_raw"] * v["launches"] for v in hits) / n}
    return {"hbm_bytes": int(k["hbm_bytes_fetch_doubled"]), "hbm_bytes_uncorrected": int(k["hbm_bytes_raw"]),
            "source": os.path.relpath(path, REPO)}


def in_step_us(kernel, ae_form, b):
    """Average duration of the kernel family inside the running step, from the committed ``rocprofv3 --kernel-trace
    --stats`` summary of this command (profiles/r*_<net>_b<batch>_kernel_stats.csv), launch-weighted over the
    template instances -- beside the probe's "alone" time (other branches of the graph share the chip at large batches)."""
    import csv
    path = _profile(f"{ae_form.lower()}_b{b}_kernel_stats.csv")
    if path is None:
        return None
    fam = kernel.split("<")[0].split("[")[0]
    calls = total = 0
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            if name.split("<")[0].split("(")[0] == fam:
                calls += int(row["Calls"])
                total += int(row["TotalDurationNs"])
    return round(total / calls / 1e3, 2) if calls else None


def annotate(roof, ae_form, b):
    """Counter traffic and in-step duration beside every probed kernel family of a roofline block."""
    for r in roof["top_kernels"]:
        t = pmc_traffic(r["kernel"], ae_form, b)
        r["traffic"] = t["hbm_bytes"] if t else None
        if t:
            r["traffic_over_algorithmic"] = round(t["hbm_bytes"] / r["algorithmic_bytes_per_launch"], 3)
            r["traffic_source"] = t["source"]
        us = in_step_us(r["kernel"], ae_form, b)
        r["in_step_launch_us"] = us
        if us:
            r["frac_in_step"] = round(r["algorithmic_bytes_per_launch"] / (us * 1e-6) / 1e9 / roof["peak"], 5)
    lead = roof["top_kernels"][0]
    roof["traffic"] = lead["traffic"]
    roof["traffic_over_algorithmic"] = lead.get("traffic_over_algorithmic")
    roof["in_step_launch_us"] = lead["in_step_launch_us"]
    roof["frac_in_step"] = lead.get("frac_in_step")
    roof["traffic_note"] = ("traffic = FETCH_SIZE x 2 + WRITE_SIZE per launch of the kernel family, from the committed "
                            "rocprofv3 --pmc passes of this command; in_step_launch_us from its committed --kernel-trace "
                            "--stats summary (profiles/); achieved / frac use the HIP-event time of this run, kernel alone")


def cpu_calibration(ae_form):
    """port / reference steps-per-second ratio measured in the build container (oracle/calibrate.py), so that the
    port's number on this box can be read as the reference's."""
    path = os.path.join(REPO, "profiles", "cpu_calibration.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        c = json.load(f)
    return c.get(ae_form, {}).get("port_over_reference")


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start the N rank processes ourselves (one
    ``torch.distributed.run`` child, rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU,
    pass their output through and exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def quiet_gc():
    """Everything allocated so far (modules, plans, captured graphs) moves to the permanent generation: a full
    collection of the cyclic garbage collector in the middle of the timed loop walks all of it and stalls the host for
    tens of milliseconds -- longer than the queue of launched steps lasts at batch 4096 (measured in 40-step windows:
    280 steps/s with dips to 170-250 several times per second; a third as many with the collector held off)."""
    gc.collect()
    gc.freeze()


def configs2_line(args, cfg0, dev):
    """BASELINE configs[2] beside the headline: 100 k synthetic spectra, batch 4096 -- the regime where the conv
    kernels stop being launch-bound.  Same engine, same step; returns steps/s and the top kernels' roofline."""
    from rankaae_amd.engine import StepEngine
    from rankaae_amd.synthetic import make_spectra
    from rankaae_amd.dataloader import split_counts
    cfg = dict(cfg0, batch_size=4096)
    rows, b = 100000, 4096
    spec, aux, _ = make_spectra(rows, cfg["dim_in"], cfg["n_aux"], seed=0)
    n_train = split_counts(rows)[0]
    enc, dec, dis = build_models(cfg, 1234)
    eng = StepEngine(enc, dec, dis, cfg, dev, rng_mode="philox", seed=1234, use_graph=not args.no_graph)
    eng.set_data(spec[:n_train], aux[:n_train])
    full = n_train // b
    gen = torch.Generator().manual_seed(7)
    i = 0

    def one_step():
        nonlocal i
        if i % full == 0:
            eng.set_epoch(torch.randperm(n_train, generator=gen), 0.7172)
        eng.step(b, smooth=True)
        i += 1
    for _ in range(20):          # eager emission, capture, first replays -- and the clocks of a GPU that sat idle
        one_step()
    torch.cuda.synchronize()
    quiet_gc()
    windows, per = 6, 40         # six 40-step windows: min / median / max instead of one figure
    rates = []
    t_all = time.perf_counter()
    for _ in range(windows):
        t0 = time.perf_counter()
        for _ in range(per):
            one_step()
        eng.finish()                  # (a deferred smoothness tail of the last step belongs to the window)
        torch.cuda.synchronize()
        rates.append(per / (time.perf_counter() - t0))
    dt = time.perf_counter() - t_all
    steps = windows * per
    order = list(rates)
    rates.sort()
    med = 0.5 * (rates[windows // 2 - 1] + rates[windows // 2])
    out = {"workload": f"BASELINE configs[2]: {rows}x{cfg['dim_in']} synthetic spectra (train split {n_train}), batch {b}, "
                       f"ae_form={cfg['ae_form']}", "value": round(med, 2), "unit": "steps/s",
           "windows": {"n": windows, "steps_each": per, "min": round(rates[0], 2), "median": round(med, 2),
                       "max": round(rates[-1], 2), "all_windows_together": round(steps / dt, 2),
                       "in_order": [round(r, 1) for r in order]},
           "spectra_per_s": round(med * b), "ms_per_step": round(1e3 / med, 3), "steps": steps}
    if not args.no_roofline:
        out["roofline"] = eng.roofline_probe(b, HBM_PEAK_GBS, reps=2)
        annotate(out["roofline"], cfg["ae_form"], b)
    return out


def trials_line(args, cfg, dev, spec, aux, counts=(1, 4, 8), rounds=150):
    """SURVEY 8f-3 / VERDICT r2 item 3: the reference's real workload is several small INDEPENDENT trials
    (example/fix_config.yaml: ``trials: 8``).  T engines -- own weights, stream and captured graph each, as
    ``train_sc``'s thread mode runs them -- are stepped round-robin from this one host thread; reported is the
    aggregate rate of five-phase steps over all T trials.  (The headline ``value`` stays the single-trial rate.)"""
    from rankaae_amd.engine import StepEngine
    from rankaae_amd.dataloader import split_counts
    n_train = split_counts(len(spec))[0]
    b = cfg["batch_size"]
    full = n_train // b
    out = {}
    for T in counts:
        engs = []
        for t in range(T):
            enc, dec, dis = build_models(cfg, 1234 + t)
            e = StepEngine(enc, dec, dis, cfg, dev, rng_mode="philox", seed=99 + t, use_graph=not args.no_graph)
            e.set_data(spec[:n_train], aux[:n_train])
            engs.append(e)
        gen = torch.Generator().manual_seed(7)
        i = 0

        def one_round():
            nonlocal i
            for e in engs:
                if i % full == 0:
                    e.set_epoch(torch.randperm(n_train, generator=gen), 0.7172)
                e.step(b, smooth=True)
            i += 1
        for _ in range(6):
            one_round()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(rounds):
            one_round()
        for e in engs:
            e.finish()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[str(T)] = round(T * rounds / dt, 1)
        del engs
        gc.collect()
        torch.cuda.empty_cache()
    base = out[str(counts[0])]
    res = {"unit": "five-phase steps/s summed over T concurrent independent trials on ONE GPU (one engine, stream and "
                   "hipGraph per trial, one host thread)", "batch": b, "aggregate_steps_per_s": out,
           "speedup_vs_one_trial": {k: round(v / base, 2) for k, v in out.items()},
           "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)")}
    if True:
        # every kernel of the step also exists in a batched form: ONE launch sequence, gridDim.z = T
        # (rankaae_amd.trial_batch.TrialBatch); each trial bit for bit what it is alone
        from rankaae_amd.trial_batch import TrialBatch
        bout = {}
        bcfg = dict(cfg)
        if cfg["ae_form"] != "FC":
            bcfg.setdefault("tile_rows_mult", 4)     # as train_sc's batched mode runs the conv networks (raae_tile_hint)
        for T in tuple(counts) + ((16,) if 16 not in counts else ()):
            shared = TrialBatch.shared_stream(dev)
            engs = []
            for t in range(T):
                enc, dec, dis = build_models(bcfg, 1234 + t)
                e = StepEngine(enc, dec, dis, bcfg, dev, rng_mode="philox", seed=99 + t, use_graph=True, stream=shared)
                e.set_data(spec[:n_train], aux[:n_train])
                engs.append(e)
            batch = TrialBatch(engs)
            gen = torch.Generator().manual_seed(7)
            i = 0

            def one_round():
                nonlocal i
                if i % full == 0:
                    for e in engs:
                        e.set_epoch(torch.randperm(n_train, generator=gen), 0.7172)
                batch.step(b, smooth=True)
                i += 1
            for _ in range(6):
                one_round()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(rounds):
                one_round()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            bout[str(T)] = round(T * rounds / dt, 1)
            batch.release()
            for e in engs:
                e.release()
            del engs, batch
            gc.collect()
            torch.cuda.empty_cache()
        res["batched_launches"] = {"unit": "the same, the T trials stepped by ONE launch sequence with gridDim.z = T "
                                           "(TrialBatch)", "tile_rows_mult": bcfg.get("tile_rows_mult", 1),
                                   "aggregate_steps_per_s": bout,
                                   "speedup_vs_one_trial": {k: round(v / base, 2) for k, v in bout.items()}}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--ae-form", default=os.environ.get("RANKAAE_BENCH_AE_FORM", "compact"))
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rows", type=int, default=7000)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU-oracle timing (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-detail", action="store_true", help="roofline probe per block shape (tuning aid)")
    ap.add_argument("--no-epoch", action="store_true", help="skip the epoch-inclusive (validation + metrics) timing")
    ap.add_argument("--no-configs2", action="store_true", help="skip the batch-4096 / 100k-row sub-run (configs[2])")
    ap.add_argument("--no-trials", action="store_true", help="skip the concurrent-trials sub-run (T = 1 / 4 / 8 engines)")
    ap.add_argument("--trials", type=str, default="1,4,8", help="engine counts of the concurrent-trials sub-run")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override an engine tuning key (side_streams, overlap_unused_forwards, fused_blocks)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)              # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal on a one-GPU box: RANKAAE_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo (RCCL refuses
    # two ranks on one device); it exercises sharding, graph segmentation and the collectives, not the speed
    rehearsal = os.environ.get("RANKAAE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible "
                         "(RANKAAE_BENCH_REHEARSAL=1 rehearses the multi-rank path on one GPU over gloo)")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from rankaae_amd.engine import StepEngine
    from rankaae_amd.synthetic import make_spectra
    from rankaae_amd.dataloader import split_counts
    from rankaae_amd import _lib
    if hasattr(_lib.load(), "raae_debug_stamps"):
        raise SystemExit("librankaae_hip.so is a -DRAAE_STAMPS build (extra barriers and clock reads in every "
                         "kernel): rebuild with plain rankaae_amd/csrc/build.sh before benchmarking")

    cfg = dict(BASE_CFG)
    cfg.update(ae_form=args.ae_form, batch_size=args.batch)
    for kv in args.set:
        k, v = kv.split("=", 1)
        cfg[k] = json.loads(v)
    # weak scaling: the data set grows with the number of ranks, so that every rank runs the same number of
    # steps per epoch as the single-GPU run (with 7000 rows fixed, 8 ranks would reshuffle every 2 steps)
    rows = args.rows * world
    spec, aux, _ = make_spectra(rows, cfg["dim_in"], cfg["n_aux"], seed=0)
    n_train = split_counts(rows)[0]
    enc, dec, dis = build_models(cfg, 1234)
    eng = StepEngine(enc, dec, dis, cfg, dev, rng_mode="philox", seed=1234 + rank, use_graph=not args.no_graph,
                     world_size=world, rank=rank)
    # every rank holds the whole training split in HBM; with N ranks a global batch is N*batch rows and
    # rank r steps rows [r*batch, (r+1)*batch) of it (same permutation on all ranks)
    eng.set_data(spec[:n_train], aux[:n_train])
    b = args.batch
    full_batches = n_train // (b * world)
    if full_batches < 1:
        raise SystemExit("dataset too small for one global batch")
    gen = torch.Generator().manual_seed(7)
    state = {"i": 0}

    def one_step():
        if state["i"] % full_batches == 0:
            perm = torch.randperm(n_train, generator=gen)
            eng.set_epoch(perm, 0.7172, start=rank * b, stride=world * b)
        eng.step(b, smooth=True)
        state["i"] += 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 3)):     # >= 3: eager emission, graph capture, first replay
        one_step()
    quiet_gc()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    eng.finish()        # `overlap_steps`: the last step's smoothness tail runs beside the NEXT step's head -- here there is none
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    losses = eng.losses()
    finite = all(np.isfinite(v) for v in losses.values())

    if rank == 0:
        value = world * args.steps / dt
        named = "BASELINE configs[1]" if (b, args.rows) == (256, 7000) else \
            ("BASELINE configs[2]" if (b, args.rows) == (4096, 100000) else "custom workload")
        rccl_ranks = 1
        if world > 1:
            rccl_ranks = eng.graph_ar.world if eng.graph_ar is not None else dist.get_world_size()
        line = {
            "metric": f"training steps/sec (batch={b}, {cfg['dim_in']}-pt spectra)", "value": round(value, 2),
            "unit": f"steps/s (five-phase steps on {b}-row batches, summed over ranks)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{named}: {rows}x{cfg['dim_in']} synthetic spectra ({args.rows} per GPU; train split "
                                   f"{n_train}), batch {b}/GPU, ae_form={cfg['ae_form']}, nstyle={cfg['nstyle']}, "
                                   f"n_aux={cfg['n_aux']}, AdamW, 5 phases incl. smoothness",
                       "global_batch": b * world, "parallelism": f"dp{world}", "hip_graph": not args.no_graph,
                       "rng": "philox tape (device)", "rccl_ranks": rccl_ranks,
                       "collective": (None if world == 1 else ("in-graph ncclAllReduce" if eng.graph_ar is not None
                                                               else f"torch.distributed {dist.get_backend()} between graph segments"))},
            "losses_finite": finite, "last_losses": {k: round(v, 6) for k, v in losses.items()},
        }
        # the extras run on ONE GPU only: with several ranks the probe's extra training steps would enter
        # collectives the other ranks never join, and the contract asks for the CPU baseline at N = 1
        if not args.no_roofline and world == 1:
            line["roofline"] = eng.roofline_probe(b, HBM_PEAK_GBS, detail=args.roofline_detail)
            annotate(line["roofline"], cfg["ae_form"], b)
            if b < 1024:
                line["roofline"]["note"] = ("this batch is launch/latency bound (SURVEY 8d): every kernel moves <= 2.5 MB; "
                                            "the HBM-bound regime is the configs2 sub-run below")
        if not args.no_epoch and world == 1:
            line["epoch_inclusive"] = epoch_inclusive(cfg, spec, aux)
        big = not args.no_configs2 and world == 1 and (b, args.rows) == (256, 7000)
        if (big or not args.no_trials) and world == 1:
            del eng
            gc.unfreeze()
            gc.collect()
            torch.cuda.empty_cache()
        # (configs2 first: its branched graph wants its streams on different hardware queues, and HIP hands queues out
        # round-robin at stream creation -- behind the 13 engines of the trials sub-run it measured 174 steps/s instead
        # of 255-275)
        if big:
            line["configs2"] = configs2_line(args, cfg, dev)
        if not args.no_trials and world == 1 and b <= 1024:
            line["concurrent_trials"] = trials_line(args, cfg, dev, spec, aux, tuple(int(x) for x in args.trials.split(",")))
        if args.cpu_budget > 0 and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, spec, aux, args.cpu_budget)
            ratio = cpu_calibration(cfg["ae_form"])
            if ratio:
                line["cpu_baseline"]["port_over_reference"] = ratio
                line["cpu_baseline"]["reference_equivalent"] = round(line["cpu_baseline"]["value"] / ratio, 3)
            line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
