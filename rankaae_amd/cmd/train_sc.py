#!/usr/bin/env python
"""``train_sc -c <config.yaml> [-w <work_dir>]`` -- same command line, per-trial directory layout
and log files as the reference's ``sc/cmd/train_sc.py:105-157`` (``training/job_<k>/{messages.txt,
losses.csv,final.pt}``, ``main_process_message.txt``), with the training itself on the MI355X HIP
engine.  The ipyparallel engine farm (``train_sc.py:19-45``) becomes one worker process per visible GPU:
trial ``k`` runs on worker ``k mod n`` and worker ``w`` uses GPU ``w mod ngpus`` (``RANKAAE_TRIAL_WORKERS``
overrides ``n``).  Several trials share a GPU (config key ``trials_per_gpu``, default 4 when ``trials > 1``): a 256-row
step is one serial chain of ~10 us kernels that leaves most of the chip idle, and the chains of independent trials
overlap on it.  ``trial_mode: auto`` (default) picks ``batched`` for the dense networks -- the trials of a group train in
LOCKSTEP and every training step of the group is ONE launch sequence whose kernels run with ``gridDim.z = trials``
(``rankaae_amd/trial_batch.py``; 6.2x the single-trial rate at 8 trials, 10x at 16) -- and ``threads`` for the conv
networks.  ``trial_mode: threads``: the trials of a worker run in threads of ONE process -- one engine,
HIP stream and captured graph per trial, all of them feeding the same device; every trial draws from its own host
generator (seeded ``trial_seed + k``; ``trial_seed`` defaults to a draw from the global generator), so a trial's result
does not depend on what runs beside it (tests/test_trainer_gpu.py).  ``trial_mode: processes``: one worker process per
concurrent trial, as in round 2.  When launched under
``torch.distributed.run`` (WORLD_SIZE > 1) every trial instead trains data-parallel over RCCL: all ranks run the trials
one after the other, each trial is ONE training run (``Trainer`` shards every global batch over the ranks and averages
the gradients, rankaae_amd/trainer.py), and rank 0 alone writes the logs, checkpoints and ``final.pt``."""
import argparse
import logging
import os
import signal
import threading
import time


import numpy as np
import torch

from rankaae_amd import _lib
from rankaae_amd.logger import create_logger
from rankaae_amd.parameter import Parameters
from rankaae_amd.trainer import Trainer


def timeout_handler(signum, frame):
    raise Exception("Training Overtime!")


def run_training(job_number, work_dir, train_config, verbose, data_file, timeout_hours=0,
                 logger=logging.getLogger("training"), host_rng=None, init_lock=None):
    """``host_rng`` / ``init_lock``: thread mode (several trials in this process) -- the trial's own host generator, and
    the lock under which the global generator is seeded for the construction of ITS networks."""
    work_dir = f"{work_dir}/training/job_{job_number + 1}"
    os.makedirs(work_dir, exist_ok=True)
    if _is_lead_rank():
        logger = create_logger(f"subtraining_{job_number + 1}", os.path.join(work_dir, "messages.txt"))
        loss_logger = create_logger(f"losses_{job_number + 1}", os.path.join(work_dir, "losses.csv"), simple_fmt=True)
    else:       # data parallel: the trial is ONE training run on all ranks, rank 0 alone keeps the logs and files
        logger = loss_logger = _null_logger()
    ngpus = torch.cuda.device_count()
    local_id = int(os.environ.get("LOCAL_RANK", os.environ.get("SLURM_LOCALID", 0)))
    igpu = local_id % ngpus if ngpus > 0 else -1
    start = time.time()
    logger.info(f"Training started for trial {job_number + 1}.")
    if host_rng is None:
        trainer = Trainer.from_data(data_file, igpu=igpu, verbose=verbose, work_dir=work_dir,
                                    config_parameters=train_config, logger=logger, loss_logger=loss_logger)
    else:
        with init_lock:      # nn.Module constructors draw their initial weights from the GLOBAL generator
            torch.manual_seed(host_rng.initial_seed())
            trainer = Trainer.from_data(data_file, igpu=igpu, verbose=verbose, work_dir=work_dir,
                                        config_parameters=train_config, logger=logger, loss_logger=loss_logger,
                                        host_rng=host_rng)
    trainer.freeze_gc = bool(train_config.get("freeze_gc", host_rng is None))   # a dedicated training process: collector held off
    timer = None
    if host_rng is not None:
        # a thread: no signals here.  A timer asks the trainer to stop; train() raises the reference's exception at the
        # next epoch boundary
        if timeout_hours > 0:
            timer = threading.Timer(float(timeout_hours) * 3600.0, trainer.request_stop, args=("Training Overtime!",))
            timer.daemon = True
            timer.start()
        try:
            metrics = trainer.train()
        finally:
            if timer is not None:
                timer.cancel()
            trainer.engine.release()          # graphs, workspaces and streams back now: more trials follow in this process
        logger.info(metrics)
        time_used = time.time() - start
        logger.info(f"Training finished. Time used: {time_used:.2f}s.\n\n")
        return metrics, time_used
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # data parallel: the rank whose alarm fires must not leave the others waiting in a collective -- it asks the
        # trainer to stop, and every rank raises the reference's exception together at the next epoch boundary
        signal.signal(signal.SIGALRM, lambda signum, frame: trainer.request_stop("Training Overtime!"))
    else:
        signal.signal(signal.SIGALRM, timeout_handler)
    # (signal.setitimer: fractional hours work too -- signal.alarm(int(...)) of the reference rounds a sub-second
    # timeout to "never"; whole seconds behave identically)
    signal.setitimer(signal.ITIMER_REAL, max(0.0, float(timeout_hours) * 3600.0))
    try:
        metrics = trainer.train()
    finally:
        signal.setitimer(signal.ITIMER_REAL, 0.0)
        trainer.engine.release()              # graphs, workspaces and streams back now: more trials may follow
    logger.info(metrics)
    time_used = time.time() - start
    logger.info(f"Training finished. Time used: {time_used:.2f}s.\n\n")
    return metrics, time_used


def _is_lead_rank():
    return int(os.environ.get("WORLD_SIZE", "1")) == 1 or int(os.environ.get("RANK", "0")) == 0


def _null_logger():
    lg = logging.getLogger("rankaae_amd.null")
    if not lg.handlers:
        lg.addHandler(logging.NullHandler())
    lg.propagate = False
    return lg


def assign_trials(trials, nworkers):
    """Trial numbers of each worker: round-robin, like the reference's ipyparallel ``map_sync`` over engines that
    were given GPU ``id % ngpus`` (``train_sc.py:32-41``)."""
    return [list(range(w, trials, nworkers)) for w in range(nworkers)]


def _trial_worker(worker, jobs, work_dir, config_dict, verbose, data_file, timeout, threads=1, trial_seed=None, batched=False):
    # a spawned process: nothing has touched the GPU yet; LOCAL_RANK picks it in run_training
    os.environ["LOCAL_RANK"] = str(worker)
    cfg = Parameters(config_dict)
    if batched:
        return run_trials_batched(jobs, threads, work_dir, cfg, verbose, data_file, timeout, trial_seed, auto=batched == 2)
    if threads <= 1:
        return [(k,) + tuple(run_training(k, work_dir, cfg, verbose, data_file, timeout)) for k in jobs]
    return run_trials_threaded(jobs, threads, work_dir, cfg, verbose, data_file, timeout, trial_seed)


def run_trials_batched(jobs, per_batch, work_dir, train_config, verbose, data_file, timeout, trial_seed, auto=False):
    """The trials ``jobs`` of this process, ``per_batch`` at a time, each group trained in LOCKSTEP: every training step
    of the group is one launch sequence with ``gridDim.z = trials`` (``rankaae_amd.trainer.train_trials_batched``; dense
    networks).  Seeds, files and log lines per trial as in the thread mode: ``[(k, metrics, time_used)]``."""
    from rankaae_amd.trainer import train_trials_batched
    plain_config = train_config
    if train_config.get("ae_form", None) != "FC" and train_config.get("tile_rows_mult", None) is None:
        # conv networks: every trial's launches sized as its share of a 4x larger batch (raae_tile_hint; +33 % at 8
        # trials, +48 % at 16).  Keyed on the trial MODE, not on the group size: a trial's result is the same whatever
        # runs beside it
        train_config = Parameters({**train_config.to_dict(), "tile_rows_mult": 4})
    ngpus = torch.cuda.device_count()
    local_id = int(os.environ.get("LOCAL_RANK", os.environ.get("SLURM_LOCALID", 0)))
    igpu = local_id % ngpus if ngpus > 0 else -1
    out = []
    for i in range(0, len(jobs), per_batch):
        group = jobs[i:i + per_batch]
        torch.cuda.set_device(max(igpu, 0))
        stream = torch.cuda.Stream()
        trainers, loggers = [], []
        start = time.time()
        for k in group:
            wd = f"{work_dir}/training/job_{k + 1}"
            os.makedirs(wd, exist_ok=True)
            logger = create_logger(f"subtraining_{k + 1}", os.path.join(wd, "messages.txt"))
            loss_logger = create_logger(f"losses_{k + 1}", os.path.join(wd, "losses.csv"), simple_fmt=True)
            logger.info(f"Training started for trial {k + 1}.")
            g = torch.Generator()
            g.manual_seed(int(trial_seed) + k)
            torch.manual_seed(g.initial_seed())          # the networks' initial weights come from the global generator
            trainers.append(Trainer.from_data(data_file, igpu=igpu, verbose=verbose, work_dir=wd,
                                              config_parameters=train_config, logger=logger, loss_logger=loss_logger,
                                              host_rng=g, engine_stream=stream))
            loggers.append(logger)
        timer = None
        if timeout > 0:
            timer = threading.Timer(float(timeout) * 3600.0, lambda: [t.request_stop("Training Overtime!") for t in trainers])
            timer.daemon = True
            timer.start()
        refused = None
        try:
            metrics = train_trials_batched(trainers)
        except (_lib.HipCallError, ValueError) as exc:
            # a step of this configuration meets a kernel without the batched form (a per-layer fallback, an unusual
            # shape): `trial_mode: batched` says so; `auto` trains the group in threads instead, from the start
            if not auto:
                raise
            refused = exc
        finally:
            if timer is not None:
                timer.cancel()
            for t in trainers:
                t.engine.release()
        if refused is not None:
            logging.getLogger("Main").warning(f"trial_mode auto: batched launches refused ({refused}); trials "
                                              f"{[k + 1 for k in group]} run in threads")
            for lg in loggers:
                for h in list(lg.handlers):
                    h.close()
                    lg.removeHandler(h)
            del trainers
            out += run_trials_threaded(group, min(4, len(group)), work_dir, plain_config, verbose, data_file, timeout, trial_seed)
            continue
        time_used = time.time() - start
        for k, m, logger in zip(group, metrics, loggers):
            logger.info(m)
            logger.info(f"Training finished. Time used: {time_used:.2f}s.\n\n")
            out.append((k, m, time_used))
    return out


def run_trials_threaded(jobs, threads, work_dir, train_config, verbose, data_file, timeout, trial_seed):
    """The trials ``jobs`` of this process, ``threads`` at a time, each in a thread with its own engine / stream / graph
    and its own host generator (seed ``trial_seed + k``): ``[(k, metrics, time_used)]``."""
    from concurrent.futures import ThreadPoolExecutor
    if train_config.get("rng_mode", "philox") != "philox":
        raise ValueError("trial_mode: threads needs rng_mode: philox (the parity mode draws every random number of "
                         "every trial from the one global CPU generator: use trial_mode: processes)")
    init_lock = threading.Lock()

    def one(k):
        g = torch.Generator()
        g.manual_seed(int(trial_seed) + k)
        return (k,) + tuple(run_training(k, work_dir, train_config, verbose, data_file, timeout, host_rng=g,
                                         init_lock=init_lock))
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(one, jobs))


def run_trials(trials, work_dir, train_config, verbose, data_file, timeout, logger):
    """All trials; returns ``[(metrics, time_used)]`` in trial order and the number of worker processes."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    nworkers = 1
    mode = str(train_config.get("trial_mode", "auto"))
    if mode not in ("auto", "batched", "threads", "processes"):
        raise ValueError(f"trial_mode must be 'auto', 'batched', 'threads' or 'processes', not {mode!r}")
    philox = train_config.get("rng_mode", "philox") == "philox"
    dense = train_config.get("ae_form", None) == "FC"
    can_batch = (philox and train_config.get("precision", "fp32") == "fp32" and
                 train_config.get("fused_step_begin", True) and train_config.get("fused_discriminator", True) and
                 (dense or int(train_config.get("batch_size", 0)) < 1024))
    if mode == "batched" and not can_batch:
        raise ValueError("trial_mode: batched needs rng_mode: philox, precision: fp32 and, for the conv networks, "
                         "batch_size < 1024 (the large-batch conv kernels have no batched form: use trial_mode: threads)")
    auto = mode == "auto"
    if auto:                # one launch sequence for all trials of a group where the step's kernels have the batched form
        mode = "batched" if can_batch else "threads"
    batched = (2 if auto else 1) if mode == "batched" else 0
    if batched:
        mode = "threads"          # same process / seed plumbing below; the worker trains its group in lockstep instead
    seeded = train_config.get("trial_seed", None) is not None
    if world == 1 and (trials > 1 or seeded) and mode == "threads" and philox:
        # one worker process per GPU; inside it `trials_per_gpu` trials at a time: in threads, or (dense networks) as
        # one batched launch sequence -- eight trials of a 256-row dense step still fit the chip side by side
        per_gpu = int(os.environ.get("RANKAAE_TRIALS_PER_GPU", train_config.get("trials_per_gpu", 8 if batched else 4)))
        ngpu = max(1, torch.cuda.device_count())
        nproc = max(1, min(int(os.environ.get("RANKAAE_TRIAL_WORKERS", ngpu)), trials))
        threads = max(1, min(per_gpu, -(-trials // nproc)))
        seed = train_config.get("trial_seed", None)
        if seed is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item()) & 0x3fffffff
        if nproc == 1:
            if batched:
                done = sorted(run_trials_batched(list(range(trials)), threads, work_dir, train_config, verbose, data_file,
                                                 timeout, seed, auto=batched == 2))
            else:
                done = sorted(run_trials_threaded(list(range(trials)), threads, work_dir, train_config, verbose, data_file,
                                                  timeout, seed))
            return [(m, t) for _, m, t in done], threads
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(nproc) as pool:
            parts = pool.starmap(_trial_worker, [(w, jobs, work_dir, train_config.to_dict(), verbose, data_file, timeout,
                                                  threads, seed, batched) for w, jobs in enumerate(assign_trials(trials, nproc))])
        done = sorted(r for part in parts for r in part)
        return [(m, t) for _, m, t in done], nproc * threads
    if world == 1 and trials > 1:
        # device_count() does not initialise the GPU, so the workers can still be spawned after it
        # `trials_per_gpu` (config key, or RANKAAE_TRIALS_PER_GPU): worker processes that share one GPU.  At batches
        # below 2048 rows a trial's step is one serial chain of small kernels, and such chains of different
        # processes overlap on the chip: measured on one MI355X at B=256, conv networks: 727 steps/s alone,
        # 2 x 657, 4 x 509 (2035 in total), 6 workers 2225 in total.
        per_gpu = int(os.environ.get("RANKAAE_TRIALS_PER_GPU", train_config.get("trials_per_gpu", 1)))
        nworkers = int(os.environ.get("RANKAAE_TRIAL_WORKERS", torch.cuda.device_count() * max(1, per_gpu)))
        nworkers = max(1, min(nworkers, trials))
    if nworkers == 1:
        return [run_training(k, work_dir, train_config, verbose, data_file, timeout, logger) for k in range(trials)], 1
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(nworkers) as pool:
        parts = pool.starmap(_trial_worker, [(w, jobs, work_dir, train_config.to_dict(), verbose, data_file, timeout)
                                             for w, jobs in enumerate(assign_trials(trials, nworkers))])
    done = sorted(r for part in parts for r in part)
    return [(m, t) for _, m, t in done], nworkers


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, required=True, help="Config for training parameter in YAML format")
    parser.add_argument("-w", "--work_dir", type=str, default=".", help="Working directory to write the output files")
    args = parser.parse_args()
    work_dir = os.path.abspath(os.path.expanduser(args.work_dir))
    train_config = Parameters.from_yaml(os.path.join(work_dir, args.config))
    assert os.path.exists(work_dir)
    verbose = train_config.get("verbose", False)
    trials = train_config.get("trials", 1)
    data_file = os.path.join(work_dir, train_config.get("data_file", None))
    timeout = train_config.get("timeout", 10)
    logger = create_logger("Main training:", f"{work_dir}/main_process_message.txt", append=True) if _is_lead_rank() \
        else _null_logger()
    logger.info("START")
    start = time.time()
    result, nworkers = run_trials(trials, work_dir, train_config, verbose, data_file, timeout, logger)
    logger.info("Running with {} process(es).".format(max(nworkers, int(os.environ.get("WORLD_SIZE", "1")))))
    time_trials = np.array([r[1] for r in result])
    logger.info(f"Time used for each trial: {time_trials.mean():.2f} +/- {time_trials.std():.2f}s.\n" +
                " ".join([f"{t:.2f}s" for t in time_trials]))
    end = time.time()
    logger.info(f"Total time used: {end - start:.2f}s for {trials} trails ({(end - start) / trials:.2f} each on average).")
    logger.info("END\n\n")


if __name__ == "__main__":
    main()
