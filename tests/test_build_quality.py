"""Static checks of the compiled kernels (hipcc cross-compiles for gfx950 without a GPU)."""
import glob
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rankaae_amd", "csrc")
# kernels allowed to touch scratch memory: a few spilled registers in instances the bench workloads never
# launch (the unsplit widest dense backward tile -- the 256- and 512-point first layers run as 128-column slices,
# instance <8, 2>; the masked variants of the large-batch conv, whose mask registers push them over the 128-VGPR budget
# of four waves per SIMD).  dense_fwd2_kernel<64, 4>: two by-value argument blocks overflow the scalar registers; the
# compiler reserves a 20-byte frame for the SGPR spill bookkeeping but the ISA holds no scratch instruction (checked
# with -save-temps).
ALLOW = {r"dense_fwd2_kernelILi64ELi4E": 32, r"dense_bwd_kernel(_m)?ILi32ELi8E": 64, r"conv_fwd_strip_kernelI.*Lb1E": 32}


def _usage(src):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", src, "-o", os.devnull,
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=CSRC)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
    assert names and len(names) == len(scratch)
    return dict(zip(names, scratch))


def test_kernels_do_not_use_scratch_memory():
    """A runtime-indexed private array silently moves to scratch memory: every access becomes an HBM round
    trip (the weight-gradient kernel once wrote 14 MB of it per launch).  No kernel may use scratch except the
    listed few-byte spills of instances outside the benchmarked shapes."""
    srcs = sorted(glob.glob(os.path.join(CSRC, "raae_*.hip")))
    with ThreadPoolExecutor(max_workers=4) as ex:
        usages = list(ex.map(_usage, srcs))
    bad = {}
    for u in usages:
        for name, nbytes in u.items():
            limit = max((v for k, v in ALLOW.items() if re.search(k, name)), default=0)
            if nbytes > limit:
                bad[name] = nbytes
    assert not bad, f"kernels using scratch memory (bytes/lane): {bad}"
