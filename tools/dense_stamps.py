"""debug: mean per-stage time of workgroup 0 of the dense kernels over a few training steps of the dense networks
(needs a -DRAAE_STAMPS build: RAAE_EXTRA_FLAGS=-DRAAE_STAMPS bash rankaae_amd/csrc/build.sh).
    python tools/dense_stamps.py [batch]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rankaae_amd.engine import StepEngine
from rankaae_amd.synthetic import make_spectra
from rankaae_amd import _lib
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows = 7000 if b <= 1024 else 100000
cfg = dict(bench.BASE_CFG); cfg.update(ae_form="FC", batch_size=b)
spec, aux, _ = make_spectra(rows, 256, cfg["n_aux"], seed=0)
n_train = int(rows * 0.7)
enc, dec, dis = bench.build_models(cfg, 1234)
eng = StepEngine(enc, dec, dis, cfg, torch.device("cuda:0"), rng_mode="philox", seed=1, use_graph=True)
eng.set_data(spec[:n_train], aux[:n_train])
eng.set_epoch(torch.randperm(n_train), 0.7)
for _ in range(12):
    eng.step(b, smooth=True)
torch.cuda.synchronize()
lib = _lib.load()
lib.raae_debug_dense_stamps.restype = ctypes.c_int
ssum = np.zeros((8, 12), dtype=np.uint64); scnt = np.zeros((8, 12), dtype=np.uint64)
assert lib.raae_debug_dense_stamps(ssum.ctypes.data_as(ctypes.c_void_p), scnt.ctypes.data_as(ctypes.c_void_p)) == 0
names = ["fwd KQ=4", "fwd KQ=16", "fwd KQ=64", "fwd KQ=128", "bwd TPW=1", "bwd TPW=4", "bwd TPW=8", "bwd TPW>=16"]
stages = {True: ["", "loads issued + statistics + slopes", "tile -> LDS", "MFMA", "epilogue (all tiles)", "partials"],
          False: ["", "W regs + loads issued + statistics", "G / X tile -> LDS", "dW MFMA", "dx MFMA + store", "slabs"]}
for k in range(8):
    if scnt[k].sum() == 0:
        continue
    n = max(int(scnt[k][1]), 1)
    print(f"{names[k]}: {n} launches, mean in-kernel {float(ssum[k].sum()) * 0.01 / n:.2f} us: " +
          "; ".join(f"{stages[k < 4][i]} {float(ssum[k][i]) * 0.01 / n:.2f}" for i in range(1, 6) if scnt[k][i]))
