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
