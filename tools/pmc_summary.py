#!/usr/bin/env python
"""Summaries of rocprofv3 ``--pmc`` passes (rocpd databases written by tools/pmc.sh).

  pmc_summary.py traffic <fetch.db> <write.db> <out.json> [note]   HBM bytes per launch and kernel:
        FETCH_SIZE and WRITE_SIZE come from SEPARATE passes (they do not fit one TCC pass together,
        MI355X_MICROARCH.md "rocprofv3 PMC slots").  gfx950 correction from the same guide: FETCH_SIZE tallies the
        128-byte requests of 16-byte-per-lane streaming reads at 64 B, so it is doubled for such kernels; other access
        widths are uncalibrated -- both the raw and the fetch-doubled figure are given.  WRITE_SIZE is exact.
  pmc_summary.py counters <pass.db> <out.json> [note]               per-kernel averages of every counter of one pass
        (e.g. SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES), plus mfma_busy_frac.
"""
import json
import re
import sqlite3
import sys


def norm(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)


def per_kernel(path):
    """``{kernel: {counter: (launches, mean value)}}``"""
    db = sqlite3.connect(path)
    out = {}
    for name, ctr, n, mean in db.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events "
                                         "group by name, counter_name"):
        out.setdefault(norm(name), {})[ctr] = (n, mean)
    return out


def traffic(fetch_db, write_db, out, note=""):
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    kernels = {}
    for k in sorted(set(f) | set(w)):
        n, fk = f.get(k, {}).get("FETCH_SIZE", (0, 0.0))
        _, wk = w.get(k, {}).get("WRITE_SIZE", (0, 0.0))
        kernels[k] = {"launches": n, "fetch_kb": round(fk, 1), "write_kb": round(wk, 1),
                      "hbm_bytes_raw": int((fk + wk) * 1024), "hbm_bytes_fetch_doubled": int((2 * fk + wk) * 1024)}
    # families (template instances merged, launch-weighted)
    fam = {}
    for k, v in kernels.items():
        g = fam.setdefault(re.sub(r"<.*", "", k), {"launches": 0, "f": 0.0, "w": 0.0})
        g["launches"] += v["launches"]
        g["f"] += v["fetch_kb"] * v["launches"]
        g["w"] += v["write_kb"] * v["launches"]
    families = {k: {"launches": g["launches"], "fetch_kb": round(g["f"] / max(g["launches"], 1), 1),
                    "write_kb": round(g["w"] / max(g["launches"], 1), 1),
                    "hbm_bytes_raw": int((g["f"] + g["w"]) / max(g["launches"], 1) * 1024),
                    "hbm_bytes_fetch_doubled": int((2 * g["f"] + g["w"]) / max(g["launches"], 1) * 1024)}
                for k, g in fam.items() if g["launches"]}
    kernels.update({k: v for k, v in families.items() if k not in kernels})
    json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate passes, tools/pmc.sh) -- "
                          "python3 bench.py ... " + note,
               "unit": "KB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE averaged over the launches of a kernel); families = "
                       "all template instances of a kernel, launch-weighted",
               "gfx950_note": "MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B for 16-B-per-lane streaming "
                              "reads (double it); 4-B-per-lane reads are uncalibrated, so raw and fetch-doubled figures are "
                              "both given; WRITE_SIZE is exact",
               "kernels": kernels}, open(out, "w"), indent=1)


def counters(path, out, note=""):
    k = per_kernel(path)
    res = {}
    for name, c in k.items():
        row = {"launches": max(n for n, _ in c.values())}
        row.update({ctr: round(v, 1) for ctr, (_, v) in c.items()})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CYCLES", (0, 0))[1] > 0:
            row["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / c["SQ_BUSY_CYCLES"][1], 4)
        res[name] = row
    json.dump({"command": "rocprofv3 --kernel-trace --pmc <counters> (tools/pmc.sh) -- python3 bench.py ... " + note,
               "unit": "counter value per launch, averaged over the launches of a kernel",
               "kernels": res}, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:6])
    else:
        counters(*sys.argv[2:5])
