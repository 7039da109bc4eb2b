"""Rank loss at validation scale (VERDICT task 9): raae_rank_loss_fwd_bwd timings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rankaae_amd import ops, _lib
_lib.load()
dev = torch.device("cuda:0")
for n in (256, 1050, 4096, 15000, 150000):
    k = 5
    g = torch.Generator().manual_seed(n)
    d = torch.randn(n, k, generator=g); d[:, 1] = torch.randint(4, 7, (n,), generator=g).float()
    z = torch.randn(n, 6, generator=g)
    d, z = d.to(dev), z.to(dev)
    work = torch.empty(ops.rank_loss_work_bytes(n, k), dtype=torch.uint8, device=dev)
    loss = torch.zeros(1, device=dev); dz = torch.empty(n, 6, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            ops.rank_loss_fwd_bwd(d, k, z, 6, n, k, True, work, loss, dz)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if n <= 15000 else 3
        e0.record(s)
        for _ in range(reps):
            ops.rank_loss_fwd_bwd(d, k, z, 6, n, k, True, work, loss, dz)
        e1.record(s)
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    print(f"n={n:7d}  {us:10.1f} us per call  {n * n * k / us / 1e6:8.1f} T pair-ops/s  loss {float(loss):.6f}")
