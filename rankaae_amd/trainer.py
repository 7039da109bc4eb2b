"""``Trainer`` -- drop-in for the reference's ``sc/clustering/trainer.py:33-474`` on MI355X.

Same constructor / ``from_data`` / ``train(callback)`` surface, same files written
(``losses.csv`` rows, ``final.pt`` / ``best.pt`` / ``checkpoints/`` whole-module pickles with the
reference's dict keys), same per-epoch validation, metrics, model selection and
``ReduceLROnPlateau`` schedule.  Everything from the batch loop downward (trainer.py:103-268)
runs as HIP kernels through ``rankaae_amd.engine.StepEngine``; there is no PyTorch/CPU fallback.

Build-only config keys (all optional, defaults preserve reference behaviour):
  ``rng_mode``  "philox" (device Philox tape, default) | "host" (reference-order CPU draws; parity)
  ``use_graph`` True: replay one captured hipGraph per step
  ``seed``      Philox seed (``rng_mode: philox``); default: drawn from the global torch generator, i.e. a different
                random stream for every trial of a run, as in the reference (whose trials are independent draws)

Data parallel (the north star's replacement of the ipyparallel trial farm): started under
``torch.distributed.run`` (WORLD_SIZE > 1) every rank builds the same ``Trainer``; rank r steps rows
``[r*b, (r+1)*b)`` of each global batch of ``W*b`` rows of the SAME epoch permutation (rank 0's), the five per-phase
gradient arenas are averaged over RCCL inside the step, BatchNorm running statistics are averaged over the ranks
before each validation, the metrics list is rank 0's on every rank (so the five ReduceLROnPlateau schedules stay in
lockstep), and only rank 0 writes ``losses.csv`` / checkpoints / ``final.pt``.
"""
import copy
import logging
import gc
import os
import shutil

import numpy as np
import torch

from .dataloader import get_dataloaders
from .engine import StepEngine
from .model import AE_CLS_DICT, DiscriminatorFC
from .parameter import OPTIM_NAMES, Parameters


def alpha(epoch_percentage, step=800, limit=0.7):
    """Gradient-reversal ramp (reference ``sc/utils/functions.py:214-219``)."""
    return (2. / (1. + np.exp(-1.0E4 / step * epoch_percentage)) - 1) * limit


class PlateauScheduler:
    """``ReduceLROnPlateau(mode="min", threshold_mode="rel", cooldown=0, min_lr=0, eps=1e-8)`` as
    the reference configures it (trainer.py:400-408), acting on one engine optimizer."""

    def __init__(self, opt, factor, patience, threshold=0.01):
        self.opt, self.factor, self.patience, self.threshold = opt, factor, patience, threshold
        self.best, self.num_bad_epochs = float("inf"), 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad_epochs = metric, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            new_lr = max(self.opt.lr * self.factor, 0.0)
            if self.opt.lr - new_lr > 1e-8:
                self.opt.lr = new_lr
                self.opt.push()
            self.num_bad_epochs = 0


class Trainer:
    metric_weights = [1.0, -1.0, -0.01, -1.0, -1.0]
    gau_kernel_size = 17

    def __init__(self, encoder, decoder, discriminator, device, train_loader, val_loader, verbose=True,
                 work_dir='.', tb_logdir="runs", config_parameters=Parameters({}),
                 logger=logging.getLogger("training"), loss_logger=logging.getLogger("losses"), engine_stream=None):
        self.logger, self.loss_logger = logger, loss_logger
        self.device = device
        self.encoder, self.decoder, self.discriminator = encoder, decoder, discriminator
        self.train_loader, self.val_loader = train_loader, val_loader
        self.verbose, self.work_dir, self.tb_logdir = verbose, work_dir, tb_logdir
        self.epoch_stop_smooth = 500
        # build-only key `freeze_gc` (default off for library use; `train_sc` and bench.py turn it on): after the second
        # epoch the long-lived objects of the run move to the cyclic collector's permanent generation for the rest of
        # train(), and come back out in a `finally` (process-global state: an embedding application opts in)
        self.freeze_gc = False
        self.__dict__.update(config_parameters.to_dict())
        if not self.gradient_reversal or self.use_cnn_discriminator:
            raise ValueError("only gradient_reversal: true with DiscriminatorFC is reachable in the reference "
                             "(SURVEY.md finding 4)")
        if self.optimizer_name not in OPTIM_NAMES:
            raise ValueError(f"optimizer_name must be one of {OPTIM_NAMES}")
        cfg = config_parameters.to_dict()
        self.world, self.rank, self.pg = self._data_parallel_setup(device)
        # one draw from the global generator per trial (also when `seed` is given, so that the generator's state
        # does not depend on the key): trials of one run then use different noise / dropout / latent streams
        # (the training loader's generator: the global one unless the trial was given its own -- train_sc's thread mode)
        host_rng = getattr(train_loader, "generator", None)
        drawn = int(torch.empty((), dtype=torch.int64).random_(generator=host_rng).item()) & 0x7fffffff if cfg.get("rng_mode", "philox") == "philox" else 0
        seed = int(cfg.get("seed", drawn)) + self.rank
        self.engine = StepEngine(encoder, decoder, discriminator, cfg, device,
                                 rng_mode=cfg.get("rng_mode", "philox"), seed=seed,
                                 use_graph=cfg.get("use_graph", True), world_size=self.world, rank=self.rank,
                                 process_group=self.pg, stream=engine_stream)
        ds = train_loader.dataset
        if ds.aux is None:
            raise ValueError("n_aux: 0 is not reachable in the reference (SURVEY.md finding 4)")
        self.engine.set_data(ds.spec, ds.aux)
        self.load_optimizers()
        self.load_schedulers()

    @staticmethod
    def _data_parallel_setup(device):
        """``(world, rank, process group)``: data parallel when started under ``torch.distributed.run``.  The process
        group is created here if the launcher's caller has not done so (backend RCCL; ``RANKAAE_DP_BACKEND=gloo``
        rehearses the path with several ranks on one GPU)."""
        import torch.distributed as dist
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world == 1 and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return 1, 0, None
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("RANKAAE_DP_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)
        return dist.get_world_size(), dist.get_rank(), None

    def load_optimizers(self):
        self.optimizers = self.engine.opts

    def load_schedulers(self):
        self.schedulers = {name: PlateauScheduler(opt, self.sch_factor, self.sch_patience)
                           for name, opt in self.optimizers.items()}

    def zerograd(self):
        """Gradients live in per-phase slabs that every backward overwrites: nothing to clear."""

    def _model_dict(self):
        """Whole-module pickles under the reference's keys (trainer.py:281-283), detached from the
        engine's arena so the file holds plain, individually-owned tensors."""
        self.engine.sync_bn_counters()
        out = {}
        for key, mod in (("Encoder", self.encoder), ("Decoder", self.decoder),
                         ("Style Discriminator", self.discriminator)):
            m = copy.deepcopy(mod)
            for p in m.parameters():
                p.data = p.data.clone()
            out[key] = m
        return out

    def train(self, callback=None):
        eng = self.engine
        try:
            # _train_epochs is a generator that hands every training step out as ("step", rows, smooth): here the
            # trial's own engine runs it; train_trials_batched runs the same request of T trials as ONE launch sequence
            gen = self._train_epochs(callback)
            try:
                req = next(gen)
                while True:
                    if req[0] == "step":
                        eng.step(req[1], smooth=req[2])
                        req = next(gen)
                    else:                                   # ("validate", spec, aux) -> (styles, losses)
                        req = gen.send(eng.validate(req[1], req[2]))
            except StopIteration as done:
                return done.value
        finally:
            # a worker process runs several trials: the next one must be able to collect this one's objects -- also when
            # the run ends in an exception (the SIGALRM "Training Overtime!", a caller's try/except around train())
            if self._gc_frozen:
                gc.unfreeze()
                self._gc_frozen = False
            # every rank reaches this point at the same place in the program: release the private RCCL communicator
            # here, not from __del__ at whatever allocation the cyclic collector happens to run
            eng.close()

    def request_stop(self, reason="Training Overtime!"):
        """Ask ``train()`` to end with ``Exception(reason)`` at the next epoch boundary -- on EVERY rank of a
        data-parallel run (the flag is OR-reduced over the ranks beside the per-epoch metrics broadcast), so that no
        rank is left waiting in a collective.  ``train_sc``'s per-trial SIGALRM calls this under data parallelism
        (reference: the handler raises at once, sc/cmd/train_sc.py:21-22; with one process that is what still happens)."""
        self._stop_reason = reason

    _stop_reason = None

    _gc_frozen = False

    def _train_epochs(self, callback):
        from .parallel import any_rank, broadcast_from_rank0, broadcast_tensor_from_rank0, epoch_schedule
        eng = self.engine
        lead = self.rank == 0
        best_combined_metric = 10.0
        chkpt_dir = f"{self.work_dir}/checkpoints"
        if lead:
            os.makedirs(chkpt_dir, exist_ok=True)
        best_chpt_file, metrics = None, None
        if lead:
            self.loss_logger.info(
                "Epoch,Train_D,Val_D,Train_G,Val_G,Train_Aux,Val_Aux,Train_Recon,"
                "Val_Recon,Train_Smooth,Val_Smooth,Train_Mutual_Info,Val_Mutual_Info")
        vds = self.val_loader.dataset
        val_spec = torch.as_tensor(vds.spec, dtype=torch.float32).contiguous().to(self.device)
        val_aux = torch.as_tensor(vds.aux, dtype=torch.float32).contiguous().to(self.device)
        n_train, bs = len(self.train_loader.dataset), self.train_loader.batch_size
        # single GPU: every batch of the reference's loader incl. the ragged last one; data parallel: every full
        # global batch of world * bs rows, the tail split evenly over the ranks (parallel.epoch_schedule)
        schedule = epoch_schedule(n_train, self.world, bs)
        n_batch = len(schedule)
        bn_buffers = [b_ for mod in (self.encoder, self.decoder) for name, b_ in mod.named_buffers()
                      if name.endswith("running_mean") or name.endswith("running_var")]
        for epoch in range(self.max_epoch):
            alpha_ = alpha(epoch / self.max_epoch, self.alpha_flat_step, self.alpha_limit)
            perm = broadcast_tensor_from_rank0(self.train_loader.epoch_permutation(), self.device, self.pg)
            eng.set_epoch(perm, alpha_)
            smooth = epoch < self.epoch_stop_smooth
            prev_rows = None
            for rows, off, global_rows in schedule:
                if self.world > 1 and rows != prev_rows:      # first step of the epoch, and again at the tail
                    eng.seek(off + self.rank * rows, global_rows)
                prev_rows = rows
                yield ("step", rows, smooth)
            if epoch == 1 and self.freeze_gc and not self._gc_frozen:
                # plans, captured graphs and modules are long-lived: out of the cyclic collector's way (a full
                # collection in the middle of an epoch stalls the host for longer than the queue of launched steps lasts)
                gc.collect()
                gc.freeze()
                self._gc_frozen = True
            tl = eng.losses()
            if not smooth:
                tl["smooth"] = 0.0
            if self.world > 1:
                eng.average_over_ranks(bn_buffers)
            z, vl = yield ("validate", val_spec, val_aux)
            if epoch % 10 == 0 and lead:
                self.loss_logger.info(
                    f"{epoch:d},\t"
                    f"{tl['adversarial']:.6f},\t{vl['adversarial']:.6f},\t"
                    f"{0.0:.6f},\t{0.0:.6f},\t"
                    f"{tl['kendall']:.6f},\t{vl['kendall']:.6f},\t"
                    f"{tl['recon']:.6f},\t{vl['recon']:.6f},\t"
                    f"{tl['smooth']:.6f},\t{vl['smooth']:.6f},\t"
                    f"{tl['mutual_info']:.6f},\t{vl['mutual_info']:.6f},\t")
            avg_mutual_info = tl["mi_accum"] / n_batch
            # shapiro(x).statistic per style / spearmanr(.).correlation per pair of the reference
            # (trainer.py:286-292), formed on the device by raae_style_metrics
            style_shapiro, style_rho = eng.val_style_metrics()
            style_coupling = np.max(np.fabs(style_rho))
            metrics = [float(np.min(style_shapiro)), vl["recon"], avg_mutual_info, float(style_coupling),
                       vl["kendall"]]
            metrics = broadcast_from_rank0(metrics, self.device, self.pg)     # no-op on one GPU
            if self.world > 1 and any_rank(self._stop_reason is not None, self.device, self.pg):
                # a rank's timeout fired: all ranks leave here together (a rank raising on its own would leave the
                # others waiting in the next step's all-reduce)
                raise Exception(self._stop_reason or "Training Overtime!")
            if self.world == 1 and self._stop_reason is not None:      # thread mode of train_sc: a timer asked for it
                raise Exception(self._stop_reason)
            combined_metric = -(np.array(self.metric_weights) * np.array(metrics)).sum()
            if combined_metric > best_combined_metric:
                best_combined_metric = combined_metric
                best_chpt_file = f"{chkpt_dir}/epoch_{epoch:06d}_loss_{combined_metric:07.6g}.pt"
                if lead:
                    torch.save(self._model_dict(), best_chpt_file)
            for sch in self.schedulers.values():
                sch.step(combined_metric)
            if callback is not None:
                callback(epoch, metrics)
        if lead:
            torch.save(self._model_dict(), f"{self.work_dir}/final.pt")
            if best_chpt_file is not None:
                shutil.copy2(best_chpt_file, f"{self.work_dir}/best.pt")
        return metrics

    @classmethod
    def from_data(cls, csv_fn, igpu=0, verbose=True, work_dir='.', train_ratio=0.7, validation_ratio=0.15,
                  test_ratio=0.15, config_parameters=Parameters({}), logger=logging.getLogger("from_data"),
                  loss_logger=logging.getLogger("losses"), arrays=None, host_rng=None, engine_stream=None):
        """``host_rng`` (build-only): a private ``torch.Generator`` for this trial's epoch permutations and device-RNG
        seed instead of the global CPU generator (several trials in one process).  ``engine_stream``: the HIP stream
        the engine works on (the trials of a ``TrialBatch`` share one)."""
        p = config_parameters
        assert p.ae_form in AE_CLS_DICT
        dl_train, dl_val, _ = get_dataloaders(csv_fn, p.batch_size, (train_ratio, validation_ratio, test_ratio),
                                              n_aux=p.n_aux, arrays=arrays, generator=host_rng)
        if not torch.cuda.is_available():
            raise RuntimeError("rankaae_amd needs an MI355X GPU: the training path has no CPU fallback "
                               "(the reference would log 'Use Slow CPU!' here)")
        if verbose:
            logger.info("Use GPU")
        device = torch.device(f"cuda:{max(igpu, 0)}")
        torch.cuda.set_device(device)
        encoder = AE_CLS_DICT[p.ae_form]["encoder"](nstyle=p.nstyle, dropout_rate=p.dropout_rate, dim_in=p.dim_in,
                                                    n_layers=p.n_layers)
        decoder = AE_CLS_DICT[p.ae_form]["decoder"](nstyle=p.nstyle, dropout_rate=p.dropout_rate,
                                                    last_layer_activation=p.decoder_activation, dim_out=p.dim_out,
                                                    n_layers=p.n_layers)
        if p.use_cnn_discriminator:
            raise ValueError("use_cnn_discriminator: true is broken in the reference (SURVEY.md finding 4)")
        discriminator = DiscriminatorFC(nstyle=p.nstyle, dropout_rate=p.dis_dropout_rate, noise=p.dis_noise,
                                        layers=p.FC_discriminator_layers)
        return cls(encoder, decoder, discriminator, device, dl_train, dl_val, verbose=verbose, work_dir=work_dir,
                   config_parameters=p, logger=logger, loss_logger=loss_logger, engine_stream=engine_stream)


def train_trials_batched(trainers, callbacks=None):
    """``train()`` of T trainers of ONE configuration in lockstep, every training step of the T trials as one launch
    sequence with ``gridDim.z = T`` (``rankaae_amd.trial_batch.TrialBatch``; dense networks).  Everything around the
    step -- epoch permutation, validation, metrics, schedulers, checkpoints, log rows -- is each trainer's own code,
    run trial after trial between the steps.  Returns the trials' metrics lists.  A trial that asks to stop
    (``request_stop``) ends the whole batch with the reference's exception: the trials advance together."""
    from .trial_batch import TrialBatch
    callbacks = callbacks or [None] * len(trainers)
    batch = TrialBatch([t.engine for t in trainers])
    gens = [t._train_epochs(cb) for t, cb in zip(trainers, callbacks)]
    results = [None] * len(trainers)
    try:
        reqs = []
        for i, g in enumerate(gens):
            try:
                reqs.append(next(g))
            except StopIteration as done:
                results[i] = done.value
                reqs.append(None)
        while not all(r is None for r in reqs):
            if any(r is None for r in reqs) or any(r[0] != reqs[0][0] or (r[0] == "step" and r != reqs[0]) for r in reqs):
                raise RuntimeError(f"the trials of a batch left lockstep: {[r and r[:1] for r in reqs]}")
            if reqs[0][0] == "step":
                batch.step(reqs[0][1], smooth=reqs[0][2])
                answers = [None] * len(gens)
            else:
                answers = batch.validate([r[1] for r in reqs], [r[2] for r in reqs])
            nxt = []
            for i, g in enumerate(gens):
                try:
                    nxt.append(next(g) if answers[i] is None else g.send(answers[i]))
                except StopIteration as done:
                    results[i] = done.value
                    nxt.append(None)
            reqs = nxt
        return results
    finally:
        for g in gens:
            g.close()
        for t in trainers:
            if t._gc_frozen:
                gc.unfreeze()
                t._gc_frozen = False
            t.engine.close()
        batch.release()
