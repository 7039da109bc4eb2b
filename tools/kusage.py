"""Register / LDS / occupancy table of every kernel in one source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
src = sys.argv[1]
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rankaae_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", src, "-o", os.devnull,
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:], capture_output=True, text=True, cwd=csrc)
blocks = re.split(r"remark: [^\n]*Function Name: ", out.stderr)[1:]
OCC, LDS = r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"
for b in blocks:
    name = b.split()[0]
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    m = re.search(r"\d+([a-z_0-9]+_kernel)", name)
    tmpl = re.findall(r"L[ib](\d+)E", name)
    print("%-34s %-14s vgpr %4s agpr %3s occ %2s lds %6s" % (m.group(1) if m else name[:34], ",".join(tmpl), g("VGPRs"), g("AGPRs"), g(OCC), g(LDS)))
