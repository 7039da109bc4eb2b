"""debug: per-stage timestamps of block 0 (needs a -DRAAE_STAMPS build)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rankaae_amd.engine import StepEngine
from rankaae_amd.synthetic import make_spectra
from rankaae_amd import _lib
cfg = dict(bench.BASE_CFG); cfg.update(ae_form="compact", batch_size=int(os.environ.get("STAMP_B", 256)))
spec, aux, _ = make_spectra(max(7000, 10 * cfg["batch_size"]), 256, cfg["n_aux"], seed=0)
n_train = int(len(spec) * 0.7)
enc, dec, dis = bench.build_models(cfg, 1234)
dev = torch.device("cuda:0")
eng = StepEngine(enc, dec, dis, cfg, dev, rng_mode="philox", seed=1, use_graph=True)
eng.set_data(spec[:n_train], aux[:n_train])
eng.set_epoch(torch.randperm(n_train), 0.7, start=0, stride=cfg["batch_size"])
for _ in range(6):
    eng.step(cfg["batch_size"], smooth=True)
torch.cuda.synchronize()
lib = _lib.load()
lib.raae_debug_stamps.restype = ctypes.c_int
out = np.zeros((4, 3, 16), dtype=np.int64)
assert lib.raae_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
names = sys.argv[1:] if len(sys.argv) > 1 else None
for k in range(4):
    for g in range(3):
        st = out[k, g]
        idx = [i for i in range(16) if st[i] != 0]
        if not idx:
            continue
        base = st[idx[0]]
        print(f"kernel {k} gridclass {g}: " + " ".join(f"[{i}]{(st[i]-base)*0.01:.2f}" for i in idx))

# per-stage totals over all launches of the run (ticks of 10 ns), per kernel and grid class (<=64, <=128, more
# workgroups): mean time per launch spent before each stamp
lib.raae_debug_stage_totals.restype = ctypes.c_int
ssum = np.zeros((4, 3, 16), dtype=np.uint64); scnt = np.zeros((4, 3, 16), dtype=np.uint64)
if lib.raae_debug_stage_totals(ssum.ctypes.data_as(ctypes.c_void_p), scnt.ctypes.data_as(ctypes.c_void_p)) == 0:
    for k in range(4):
        for g in range(3):
            if scnt[k, g].sum() == 0:
                continue
            launches = max(int(scnt[k, g][1]), 1)
            print(f"kernel {k} gridclass {g}: mean in-kernel {float(ssum[k, g].sum()) * 0.01 / launches:.2f} us over {launches} launches; "
                  "stage means (us/launch): " + " ".join(f"[{i}]{float(ssum[k, g][i]) * 0.01 / launches:.2f}" for i in range(1, 16) if scnt[k, g][i]))

# per block shape (grids of > 128 workgroups only): mean time per launch before each stamp
if hasattr(lib, "raae_debug_kind_totals"):
    ksum = np.zeros((8, 4, 16), dtype=np.uint64); kcnt = np.zeros((8, 4, 16), dtype=np.uint64)
    lib.raae_debug_kind_totals.restype = ctypes.c_int
    if lib.raae_debug_kind_totals(ksum.ctypes.data_as(ctypes.c_void_p), kcnt.ctypes.data_as(ctypes.c_void_p)) == 0:
        fam = ["fwd_a", "fwd_b", "bwd_b", "bwd_a"]
        for kind in range(8):
            for k in range(4):
                if kcnt[kind, k].sum() == 0:
                    continue
                launches = max(int(kcnt[kind, k][1]), 1)
                print(f"shape {kind if kind < 7 else 'generic'} {fam[k]}: {float(ksum[kind, k].sum()) * 0.01 / launches:.2f} us/launch over {launches}; stages: "
                      + " ".join(f"[{i}]{float(ksum[kind, k][i]) * 0.01 / launches:.2f}" for i in range(1, 16) if kcnt[kind, k][i]))
