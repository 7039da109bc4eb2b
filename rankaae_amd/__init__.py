"""rankaae_amd: the RankAAE training hot path on MI355X (HIP kernels behind the reference's Trainer / train_sc interface)."""
import os

# The HIP runtime copies every launch's kernel arguments into a per-queue pool of 1 MB by default and blocks the submitting
# thread when it is full.  A 4096-row step is ~180 launches with ~110 KB of arguments: the host could only get nine steps
# (~35 ms) ahead of the GPU, and any host stall longer than that -- a neighbour on the machine, a collector pass --
# starved it (40-step windows at 150-250 instead of 280 steps/s).  With 32 MB the host runs a whole
# epoch ahead.  Read by the runtime when it initialises, i.e. at the first HIP call of the process: this package is
# imported before that in bench.py, train_sc and the tests; a process that has already touched the GPU keeps its value.
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
