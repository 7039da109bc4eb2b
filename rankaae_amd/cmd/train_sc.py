#!/usr/bin/env python
"""``train_sc -c <config.yaml> [-w <work_dir>]`` -- same command line, per-trial directory layout
and log files as the reference's ``sc/cmd/train_sc.py:105-157`` (``training/job_<k>/{messages.txt,
losses.csv,final.pt}``, ``main_process_message.txt``), with the training itself on the MI355X HIP
engine.  The ipyparallel engine farm is gone: trials run back to back on this process's GPU, and
when launched under ``torch.distributed.run`` (WORLD_SIZE > 1) every trial trains data-parallel
over RCCL (``rankaae_amd.parallel``)."""
import argparse
import logging
import os
import signal
import time

import numpy as np
import torch

from rankaae_amd.logger import create_logger
from rankaae_amd.parameter import Parameters
from rankaae_amd.trainer import Trainer


def timeout_handler(signum, frame):
    raise Exception("Training Overtime!")


def run_training(job_number, work_dir, train_config, verbose, data_file, timeout_hours=0,
                 logger=logging.getLogger("training")):
    work_dir = f"{work_dir}/training/job_{job_number + 1}"
    os.makedirs(work_dir, exist_ok=True)
    logger = create_logger(f"subtraining_{job_number + 1}", os.path.join(work_dir, "messages.txt"))
    loss_logger = create_logger(f"losses_{job_number + 1}", os.path.join(work_dir, "losses.csv"), simple_fmt=True)
    ngpus = torch.cuda.device_count()
    local_id = int(os.environ.get("LOCAL_RANK", os.environ.get("SLURM_LOCALID", 0)))
    igpu = local_id % ngpus if ngpus > 0 else -1
    start = time.time()
    logger.info(f"Training started for trial {job_number + 1}.")
    trainer = Trainer.from_data(data_file, igpu=igpu, verbose=verbose, work_dir=work_dir,
                                config_parameters=train_config, logger=logger, loss_logger=loss_logger)
    signal.signal(signal.SIGALRM, timeout_handler)
    signal.alarm(int(timeout_hours * 3600))
    metrics = trainer.train()
    logger.info(metrics)
    signal.alarm(0)
    time_used = time.time() - start
    logger.info(f"Training finished. Time used: {time_used:.2f}s.\n\n")
    return metrics, time_used


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
    logger = create_logger("Main training:", f"{work_dir}/main_process_message.txt", append=True)
    logger.info("START")
    logger.info("Running with {} process(es).".format(int(os.environ.get("WORLD_SIZE", "1"))))
    start = time.time()
    result = [run_training(k, work_dir, train_config, verbose, data_file, timeout, logger) for k in range(trials)]
    time_trials = np.array([r[1] for r in result])
    logger.info(f"Time used for each trial: {time_trials.mean():.2f} +/- {time_trials.std():.2f}s.\n" +
                " ".join([f"{t:.2f}s" for t in time_trials]))
    end = time.time()
    logger.info(f"Total time used: {end - start:.2f}s for {trials} trails ({(end - start) / trials:.2f} each on average).")
    logger.info("END\n\n")


if __name__ == "__main__":
    main()
