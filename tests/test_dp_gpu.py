"""Data-parallel code path on real hardware (one GPU is all this pipeline has): the engine is told
``world_size=2`` over a one-rank RCCL process group, so everything the multi-GPU run does is exercised
-- flat-gradient slab reduction, the all-reduce call on the engine's stream, Adam on the averaged
single slab, and the hipGraph cut into segments around the five collectives -- and the result must be
BITWISE that of the plain single-GPU engine (a 1-rank mean is the identity)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case,in_graph", [("fc_small", True), ("compact_small", True), ("compact_small", False)])
def test_dp_path_one_rank_equals_plain(case, in_graph):
    import torch.distributed as dist
    import test_engine_gpu as T
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    g, cfg, spec, aux = T.load_case(case)
    cfg = dict(cfg)
    cfg["in_graph_allreduce"] = in_graph      # own RCCL communicator captured in the graph | graph cut at the collectives
    bs = cfg["batch_size"]
    results = []
    for world in (1, 2):
        torch.manual_seed(77)
        from rankaae_amd import model as pm
        from rankaae_amd.engine import StepEngine
        cls = pm.AE_CLS_DICT[cfg["ae_form"]]
        enc = cls["encoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], dim_in=cfg["dim_in"],
                             n_layers=cfg["n_layers"])
        dec = cls["decoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"],
                             last_layer_activation=cfg["decoder_activation"], dim_out=cfg["dim_out"],
                             n_layers=cfg["n_layers"])
        dis = pm.DiscriminatorFC(nstyle=cfg["nstyle"], dropout_rate=cfg["dis_dropout_rate"], noise=cfg["dis_noise"],
                                 layers=cfg["FC_discriminator_layers"])
        eng = StepEngine(enc, dec, dis, cfg, T.DEV, rng_mode="philox", seed=9, use_graph=True, world_size=world, rank=0)
        n_train = int(len(spec) * 0.7)
        eng.set_data(spec[:n_train], aux[:n_train])
        eng.set_epoch(torch.randperm(n_train, generator=torch.Generator().manual_seed(1)), 0.3)
        for _ in range(5):
            eng.step(bs)
        torch.cuda.synchronize()
        results.append((eng.arena.P.clone(), eng.losses()))
        if world == 2:
            items = eng.plans[bs].graphs[True]
            cuts = sum(1 for it in items if not hasattr(it, "launch"))
            if in_graph:
                assert eng.graph_ar is not None, "the in-graph all-reduce failed its self-test on this box"
                assert cuts == 0 and len(items) == 1, "one graph per step with the five all-reduces inside"
            else:
                assert eng.graph_ar is None and cuts == 5, "five collectives between graph segments"
    # stop RCCL's watchdog thread before other tests capture graphs in this process: it polls its events
    # with hipEventQuery, which HIP rejects process-wide while ANY stream is capturing
    dist.destroy_process_group()
    assert torch.equal(results[0][0], results[1][0])
    assert results[0][1] == results[1][1]


# ---------------------------------------------------------------------------------------------------------------------
# Two REAL ranks (two processes) on the one GPU over gloo: the N > 1 path end to end -- broadcast of the weights,
# sharded gather, per-replica BatchNorm, the rank loss over local or global pairs, the per-phase gradient mean and
# Adam on it -- against the sharded CPU emulation (oracle.ref_train.train_step_sharded, SURVEY 8e "parity under DP").
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _two_rank_worker(rank, world, port, case, pairs, use_graph):
    import numpy as np
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import test_engine_gpu as T
    from oracle import ref_train
    from rankaae_amd import model as pm
    from rankaae_amd.engine import StepEngine
    torch.set_num_threads(1)
    g, cfg, spec, aux = T.load_case(case)
    cfg = dict(cfg, rank_loss_pairs=pairs, pair_unused_forwards=False)
    b = cfg["batch_size"] // world                 # the fixture's batch is the GLOBAL batch here
    cfg["batch_size"] = b                          # z_real of the adversarial phase has the configured (per-rank) size
    seed = g["model_seed"]
    # -- the emulation: `world` replicas with identical weights, each with its own generator state
    replicas = []
    for _ in range(world):
        torch.manual_seed(seed)
        tr = ref_train.OracleTrainer(spec, aux, cfg)
        for m in (tr.encoder, tr.decoder, tr.discriminator):
            m.train()
        replicas.append(tr)
    n_train = len(replicas[0].train_spec)
    perm = torch.randperm(n_train, generator=torch.Generator().manual_seed(11))
    rows = [perm[r * b:(r + 1) * b].numpy() for r in range(world)]
    shards = [(torch.tensor(replicas[0].train_spec[rw], dtype=torch.float32),
               torch.tensor(replicas[0].train_aux[rw], dtype=torch.float32)) for rw in rows]
    states = []
    for r in range(world):
        torch.manual_seed(1000 + r)
        states.append(torch.get_rng_state())
    alpha_ = ref_train.alpha(0.3, cfg["alpha_flat_step"], cfg["alpha_limit"])
    o_local, o_mean = {}, {}

    def o_hook(name, local, mean):
        o_local[name], o_mean[name] = local[rank], mean
    # teacher forcing at phase granularity, as in P2 (tests/test_engine_gpu.py): this rank's replica after every
    # optimizer step
    o_post = {}
    mine_o = replicas[rank]
    mine_o.post_hook = lambda name: o_post.__setitem__(name, T._snapshot(mine_o, name))
    smooth = 0 < cfg.get("epoch_stop_smooth", 500)
    want = ref_train.train_step_sharded(replicas, shards, alpha_, states, epoch=0 if smooth else 10 ** 9,
                                        global_pairs=pairs == "global", hook=o_hook)[rank]
    # -- the engine, rank `rank` of `world`
    torch.manual_seed(seed)
    cls = pm.AE_CLS_DICT[cfg["ae_form"]]
    enc = cls["encoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], dim_in=cfg["dim_in"], n_layers=cfg["n_layers"])
    dec = cls["decoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], last_layer_activation=cfg["decoder_activation"],
                         dim_out=cfg["dim_out"], n_layers=cfg["n_layers"])
    dis = pm.DiscriminatorFC(nstyle=cfg["nstyle"], dropout_rate=cfg["dis_dropout_rate"], noise=cfg["dis_noise"],
                             layers=cfg["FC_discriminator_layers"])
    eng = StepEngine(enc, dec, dis, cfg, T.DEV, rng_mode="host", use_graph=use_graph, world_size=world, rank=rank)
    eng.set_data(spec[:n_train], aux[:n_train])
    members = {"adversarial": ("disc", "enc"), "correlation": ("enc",), "reconstruction": ("enc", "dec"),
               "mutual_info": ("enc", "dec"), "smoothness": ("dec",)}
    mine = {"disc": list(eng.dis_mod.parameters()), "enc": list(eng.enc_mod.parameters()), "dec": list(eng.dec_mod.parameters())}
    e_params = {n: [p for grp in mem for p in mine[grp]] for n, mem in members.items()}
    theirs = {"disc": list(mine_o.discriminator.parameters()), "enc": list(mine_o.encoder.parameters()),
              "dec": list(mine_o.decoder.parameters())}
    o_params = {n: [p for grp in mem for p in theirs[grp]] for n, mem in members.items()}
    e_local, e_mean = {}, {}
    eng.phase_hook = lambda name, P: e_local.__setitem__(name, eng.phase_gradient(P, name).cpu())

    def after(name, P):
        e_mean[name] = eng.G_flat[eng.opts[name].lo:eng.opts[name].hi].cpu().clone()
        snap = o_post[name]                         # engine <- this rank's replica right after its optimizer step
        eng.enc_mod.load_state_dict(snap["enc"])
        eng.dec_mod.load_state_dict(snap["dec"])
        eng.dis_mod.load_state_dict(snap["dis"])
        o = eng.opts[name]
        for p_e, p_o in zip(e_params[name], o_params[name]):
            st = snap["opt"][id(p_o)]
            off = eng.arena.off(p_e) - o.lo
            o.m[off:off + p_e.numel()].copy_(st["exp_avg"].reshape(-1))
            o.v[off:off + p_e.numel()].copy_(st["exp_avg_sq"].reshape(-1))
    eng.post_phase_hook = after
    init = eng.arena.P.clone()
    reps = 3 if use_graph else 1                   # graph mode: eager emission, capture, and the compared REPLAY
    for rep in range(reps):
        eng.arena.P.copy_(init)
        for o in eng.opts.values():
            o.m.zero_(); o.v.zero_()
        eng.steps_dev.zero_()
        torch.manual_seed(1000 + rank)
        eng.set_epoch(perm, alpha_, start=rank * b, stride=world * b)
        eng.step(b, smooth=smooth)
    torch.cuda.synchronize()
    got = eng.losses()
    bad = []
    for key, name in (("adversarial", "adversarial"), ("kendall", "correlation"), ("recon", "reconstruction"),
                      ("mutual_info", "mutual_info"), ("smooth", "smoothness")):
        if key == "smooth" and not smooth:
            continue
        # rank loss: a pair whose style difference is rounding residue may be counted on the other side (pair COUNTS
        # weigh the concordant pairs, functions.py:73-75): 32-row shards have ~500 pairs per descriptor
        tol_abs = 1e-4 * abs(want[key]) + 1e-6
        if key == "kendall":       # up to three such pairs, each worth 4 / (n^2 - n) of c_k (see test_p4_* for the bound)
            n_pairs_rows = world * b if pairs == "global" else b
            tol_abs += 3 * 0.25 * 4.0 / (n_pairs_rows * n_pairs_rows - n_pairs_rows)
        if abs(got[key] - want[key]) > tol_abs:
            bad.append(f"rank {rank} loss {key}: hip {got[key]!r} emulation {want[key]!r}")
    for name in members:
        if name == "smoothness" and not smooth:
            continue
        lo = eng.opts[name].lo
        for which, e_flat, o_list in (("local", e_local[name], o_local[name]), ("mean", e_mean[name], o_mean[name])):
            phase_max = max([float(t.abs().max()) for t in o_list if t is not None] + [0.0])
            for p_e, g_o in zip(e_params[name], o_list):
                off = eng.arena.off(p_e) - lo
                mine_g = e_flat[off:off + p_e.numel()].view(p_e.shape).double()
                ref_g = torch.zeros_like(mine_g) if g_o is None else g_o.double()
                err, scale = float((mine_g - ref_g).abs().max()), float(ref_g.abs().max())
                # 5e-3 |g|inf as in P2; a flipped pair count moves every rank-loss gradient (2e-2); the conv networks'
                # BatchNorms over 32-row shards amplify fp32 rounding twice as much as P2's 64-row batches (1e-2)
                tol = 2e-2 if name == "correlation" else (1e-2 if cfg["ae_form"] == "compact" else 5e-3)
                if err > tol * scale + 1e-5 * phase_max + 1e-7:
                    bad.append(f"rank {rank} {name} {which} gradient: err {err:.3e} |g|inf {scale:.3e}")
    if use_graph:
        items = eng.plans[b].graphs[bool(smooth)]
        assert items is not None and sum(1 for it in items if hasattr(it, "launch")) >= 5
    dist.barrier()
    dist.destroy_process_group()
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("case,pairs,use_graph", [("fc_small", "local", False), ("compact_small", "local", True),
                                                  ("fc_small", "global", True), ("compact_small", "global", False)])
def test_two_ranks_match_sharded_oracle(case, pairs, use_graph):
    import torch.multiprocessing as mp
    mp.spawn(_two_rank_worker, args=(2, _free_port(), case, pairs, use_graph), nprocs=2, join=True)


def _trainer_dp_worker(rank, world, port, work_dir, case):
    import json
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", RANKAAE_DP_BACKEND="gloo")
    import torch.distributed as dist
    import test_engine_gpu as T
    from rankaae_amd.parameter import Parameters
    from rankaae_amd.trainer import Trainer
    g, cfg, spec, aux = T.load_case(case)
    cfg = dict(cfg, max_epoch=3, batch_size=32, seed=5)
    torch.manual_seed(g["model_seed"] + rank)          # ranks start from DIFFERENT weights: rank 0's must win
    lines = []

    class Log:
        def info(self, msg):
            lines.append(msg)
    tr = Trainer.from_data(None, igpu=0, verbose=False, work_dir=work_dir, config_parameters=Parameters(cfg),
                           logger=Log(), loss_logger=Log(), arrays=(spec, aux))
    assert tr.world == world and tr.rank == rank and tr.engine.world_size == world
    seen = []
    metrics = tr.train(callback=lambda ep, m: seen.append(list(m)))
    torch.cuda.synchronize()
    # every rank ends with the same weights, the same metrics and the same learning rates
    flat = tr.engine.arena.P.detach().cpu()
    box = [None] * world
    dist.all_gather_object(box, (flat, seen, [o.lr for o in tr.optimizers.values()]))
    for other in box[1:]:
        assert torch.equal(box[0][0], other[0]), "ranks diverged"
        assert box[0][1] == other[1] and box[0][2] == other[2]
    assert len(seen) == 3 and all(np.isfinite(metrics))
    # only rank 0 writes: one header + one row (epoch 0) in its loss log, final.pt once
    if rank == 0:
        assert lines and lines[0].startswith("Epoch,Train_D") and lines[1].startswith("0,\t")
        assert os.path.exists(os.path.join(work_dir, "final.pt"))
    else:
        assert not lines
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_data_parallel_two_ranks(tmp_path):
    """``Trainer.from_data(...).train()`` under WORLD_SIZE = 2 (what ``torch.distributed.run -m
    rankaae_amd.cmd.train_sc`` starts): the ranks train ONE model data-parallel -- same permutation, disjoint
    shards incl. the split tail, gradient mean per phase, metrics and schedules in lockstep -- and only rank 0
    writes the log rows and ``final.pt``."""
    import torch.multiprocessing as mp
    mp.spawn(_trainer_dp_worker, args=(2, _free_port(), str(tmp_path), "compact_small"), nprocs=2, join=True)
    model = torch.load(os.path.join(str(tmp_path), "final.pt"), map_location="cpu", weights_only=False)
    assert set(model) == {"Encoder", "Decoder", "Style Discriminator"}


def test_train_sc_under_torch_distributed_run(tmp_path):
    """The documented multi-GPU command: ``python -m torch.distributed.run --nproc-per-node 2 -m
    rankaae_amd.cmd.train_sc -c cfg.yaml -w dir`` -- here two ranks on the one GPU over gloo (RANKAAE_DP_BACKEND).
    Each trial is ONE data-parallel training run: one job directory per trial with ONE set of log lines and files,
    written by rank 0 (before, every rank trained every trial on its own and raced on the files)."""
    import json
    import subprocess
    import sys
    import yaml
    from rankaae_amd.synthetic import make_spectra, write_csv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "ref_fc_small.json")) as f:
        g = json.load(f)
    cfg = dict(g["config"])
    cfg.update(max_epoch=11, trials=2, data_file="data.csv", verbose=False, timeout=1, batch_size=32)
    spec, aux, grid = make_spectra(g["n_rows"], g["n_points"], cfg["n_aux"], seed=g["data_seed"])
    write_csv(str(tmp_path / "data.csv"), spec, aux, grid)
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    env = dict(os.environ, PYTHONPATH=root, RANKAAE_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m", "rankaae_amd.cmd.train_sc",
                        "-c", "cfg.yaml", "-w", str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    main_log = (tmp_path / "main_process_message.txt").read_text()
    assert main_log.count("START") == 1 and main_log.count("END") == 1 and "Running with 2 process(es)." in main_log
    for k in (1, 2):
        job = tmp_path / "training" / f"job_{k}"
        msgs = (job / "messages.txt").read_text()
        assert msgs.count("Training started") == 1 and msgs.count("Training finished") == 1
        rows = [ln for ln in (job / "losses.csv").read_text().splitlines() if ln.strip()]
        assert rows[0].startswith("Epoch,Train_D") and [ln.split(",")[0] for ln in rows[1:]] == ["0", "10"], rows
        model = torch.load(job / "final.pt", map_location="cpu", weights_only=False)
        assert set(model) == {"Encoder", "Decoder", "Style Discriminator"}


def test_communicator_released_at_a_defined_point_not_by_the_collector():
    """the private RCCL communicator used to be released only by ``StepEngine.__del__``, i.e. by the cyclic
    collector at an arbitrary allocation -- possibly inside the NEXT trial's hipGraph capture, where ``ncclCommDestroy``
    is illegal.  Now ``Trainer.train`` ends with ``engine.close()``; ``close()`` refuses while any capture is in
    progress in the process, drops the captured steps that hold the communicator's nodes, and a second engine built
    afterwards captures its own."""
    import torch.distributed as dist
    import test_engine_gpu as T
    from rankaae_amd import ops
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29643")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    g, cfg, spec, aux = T.load_case("fc_small")
    cfg = dict(cfg, in_graph_allreduce=True)
    bs = cfg["batch_size"]

    def run(n_steps):
        from rankaae_amd import model as pm
        from rankaae_amd.engine import StepEngine
        torch.manual_seed(77)
        cls = pm.AE_CLS_DICT[cfg["ae_form"]]
        enc = cls["encoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], dim_in=cfg["dim_in"], n_layers=cfg["n_layers"])
        dec = cls["decoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], last_layer_activation=cfg["decoder_activation"],
                             dim_out=cfg["dim_out"], n_layers=cfg["n_layers"])
        dis = pm.DiscriminatorFC(nstyle=cfg["nstyle"], dropout_rate=cfg["dis_dropout_rate"], noise=cfg["dis_noise"],
                                 layers=cfg["FC_discriminator_layers"])
        eng = StepEngine(enc, dec, dis, cfg, T.DEV, rng_mode="philox", seed=9, use_graph=True, world_size=2, rank=0)
        n_train = int(len(spec) * 0.7)
        eng.set_data(spec[:n_train], aux[:n_train])
        eng.set_epoch(torch.randperm(n_train, generator=torch.Generator().manual_seed(1)), 0.3)
        for _ in range(n_steps):
            eng.step(bs)
        torch.cuda.synchronize()
        return eng
    try:
        first = run(4)
        assert first.graph_ar is not None, "the in-graph all-reduce failed its self-test on this box"
        assert first.plans[bs].graphs[True] is not None
        # inside a capture window (of any engine of the process) close() must do nothing
        ops.Graph.active += 1
        try:
            first.close()
            assert first.graph_ar is not None
        finally:
            ops.Graph.active -= 1
        ref = first.arena.P.clone()
        first.close()
        assert first.graph_ar is None and first.plans[bs].graphs == {}, "communicator and its captured steps are gone"
        second = run(4)                       # builds its own communicator and captures while `first` is still alive
        assert second.graph_ar is not None and torch.equal(second.arena.P, ref)
        # the closed engine still steps: it emits and captures again, now over torch.distributed between segments
        first.step(bs)
        first.step(bs)
        first.step(bs)
        torch.cuda.synchronize()
        assert sum(1 for it in first.plans[bs].graphs[True] if not hasattr(it, "launch")) == 5
        second.close()
    finally:
        dist.destroy_process_group()


def _stop_worker(rank, world, port, work_dir, case):
    import signal
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", RANKAAE_DP_BACKEND="gloo")
    import torch.distributed as dist
    import test_engine_gpu as T
    from rankaae_amd.parameter import Parameters
    from rankaae_amd.trainer import Trainer
    g, cfg, spec, aux = T.load_case(case)
    cfg = dict(cfg, max_epoch=100000, batch_size=32, seed=5)
    torch.manual_seed(g["model_seed"])

    class Log:
        def info(self, msg):
            pass
    tr = Trainer.from_data(None, igpu=0, verbose=False, work_dir=work_dir, config_parameters=Parameters(cfg),
                           logger=Log(), loss_logger=Log(), arrays=(spec, aux))
    seen = []
    # what rankaae_amd.cmd.train_sc.run_training installs under WORLD_SIZE > 1 -- on ONE rank only here (rank 1):
    # the other rank has no timer at all and must still come down
    if rank == 1:
        signal.signal(signal.SIGALRM, lambda signum, frame: tr.request_stop("Training Overtime!"))
        signal.setitimer(signal.ITIMER_REAL, 1.0)
    t0 = time.time()
    try:
        tr.train(callback=lambda ep, m: seen.append(ep))
        raised = None
    except Exception as e:      # noqa: BLE001 -- the reference raises a bare Exception (sc/cmd/train_sc.py:21-22)
        raised = str(e)
    elapsed = time.time() - t0
    box = [None] * world
    dist.all_gather_object(box, (raised, len(seen)))
    assert all(b_[0] == "Training Overtime!" for b_ in box), box
    # same epoch on every rank, long before the end (an epoch of this run takes 0.5-2.4 s with two ranks on one GPU: the
    # one-second timer fires during the first, second or third)
    assert box[0][1] == box[1][1] and 0 <= box[0][1] < 100000, box
    assert elapsed < 120
    assert tr.engine.graph_ar is None and not tr._gc_frozen              # train()'s finally ran
    dist.barrier()
    dist.destroy_process_group()


def test_timeout_on_one_rank_stops_every_rank(tmp_path):
    """under data parallelism the per-trial timeout (reference: SIGALRM raises "Training
    Overtime!", sc/cmd/train_sc.py:21-22,91-97) fires on ONE rank; raising there alone would leave the other rank
    waiting in the next all-reduce.  The handler now asks the trainer to stop, the flag is OR-reduced beside the
    per-epoch metrics broadcast, and both ranks raise the reference's exception after the same epoch."""
    import torch.multiprocessing as mp
    mp.spawn(_stop_worker, args=(2, _free_port(), str(tmp_path), "fc_small"), nprocs=2, join=True)


def _sharded_val_worker(rank, world, port, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import test_engine_gpu as T
    from oracle import ref_train
    from rankaae_amd import model as pm
    from rankaae_amd.engine import StepEngine
    g, cfg, spec, aux = T.load_case(case)
    cfg = dict(cfg, batch_size=cfg["batch_size"] // world)
    torch.manual_seed(g["model_seed"])
    cls = pm.AE_CLS_DICT[cfg["ae_form"]]
    enc = cls["encoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], dim_in=cfg["dim_in"], n_layers=cfg["n_layers"])
    dec = cls["decoder"](nstyle=cfg["nstyle"], dropout_rate=cfg["dropout_rate"], last_layer_activation=cfg["decoder_activation"],
                         dim_out=cfg["dim_out"], n_layers=cfg["n_layers"])
    dis = pm.DiscriminatorFC(nstyle=cfg["nstyle"], dropout_rate=cfg["dis_dropout_rate"], noise=cfg["dis_noise"],
                             layers=cfg["FC_discriminator_layers"])
    eng = StepEngine(enc, dec, dis, cfg, T.DEV, rng_mode="host", use_graph=True, world_size=world, rank=rank)
    n_train, n_val, _ = ref_train.split_rows(len(spec))
    eng.set_data(spec[:n_train], aux[:n_train])
    b = cfg["batch_size"]
    eng.set_epoch(torch.randperm(n_train, generator=torch.Generator().manual_seed(11)), 0.3, start=rank * b, stride=world * b)
    torch.manual_seed(5 + rank)
    for _ in range(2):                              # BatchNorm running statistics away from their initial values
        eng.step(b)
    bufs = [b_ for mod in (eng.enc_mod, eng.dec_mod) for n_, b_ in mod.named_buffers() if "running" in n_]
    eng.average_over_ranks(bufs)
    # an ODD number of validation rows: the two shards differ in size
    nv = n_val - (1 - n_val % 2)
    vs = torch.tensor(spec[n_train:n_train + nv], dtype=torch.float32, device=T.DEV)
    va = torch.tensor(aux[n_train:n_train + nv], dtype=torch.float32, device=T.DEV)
    out = {}
    for shard in (False, True, True):
        eng.cfg["shard_validation"] = shard
        torch.manual_seed(99)                       # same z_sample / z_real draws in every call and on every rank
        z, losses = eng.validate(vs, va)
        out.setdefault(shard, []).append((z.cpu().clone(), losses, eng.val_style_metrics()))
    rep, sh = out[False][0], out[True]
    for z, losses, (w, rho) in sh:
        assert torch.equal(z, rep[0])
        for k, v in rep[1].items():
            tol = 1e-6 * abs(v) + 1e-9
            assert abs(losses[k] - v) <= tol, (k, losses[k], v)
        assert (w == rep[2][0]).all() and (rho == rep[2][1]).all()
    # ... and identical on every rank
    box = [None] * world
    dist.all_gather_object(box, sh[-1][1])
    assert box[0] == box[1], box
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["fc_small", "compact_small"])
def test_sharded_validation_equals_replicated(case):
    """under data parallelism every rank pairs ITS rows of the validation set with all rows
    (``raae_rank_rows_pairs``), the per-descriptor totals are summed over the ranks and ``raae_rank_rows_finish`` forms
    the rank loss -- equal to the replicated ``raae_rank_loss_fwd_bwd`` over the whole set to 1e-6 (the counts are
    exact integers, the sums float64), with the other four losses, the styles and the style metrics untouched."""
    import torch.multiprocessing as mp
    mp.spawn(_sharded_val_worker, args=(2, _free_port(), case), nprocs=2, join=True)
