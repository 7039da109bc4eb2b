"""Bitwise reproducibility at the benchmark shape (B=256, compact and FC): two graph-replay runs and an
eager run from the same seed give identical weights after 4 steps -- no atomics, fixed-order partial
sums everywhere, and the multi-stream graph has every edge it needs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ae_form", ["compact", "FC"])
def test_bench_shape_bitwise_reproducible(ae_form):
    import test_engine_gpu as T
    from bench import BASE_CFG
    from rankaae_amd.synthetic import make_spectra
    cfg = dict(BASE_CFG)
    cfg.update(ae_form=ae_form, batch_size=256)
    spec, aux, _ = make_spectra(2000, 256, 5, seed=0)
    outs = []
    for use_graph in (False, True, True):
        eng = T.build_engine(cfg, 99, spec, aux, use_graph=use_graph, rng_mode="philox")
        eng.set_epoch(torch.randperm(len(eng.train_spec), generator=torch.Generator().manual_seed(0)), 0.5)
        for _ in range(4):
            eng.step(256)
        torch.cuda.synchronize()
        outs.append(eng.arena.P.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
