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
    """``{kernel: {counter: (launches, mean value per launch)}}`` -- a raw counter has one record per hardware instance
    (XCD / shader engine) of a dispatch: they are summed per dispatch first.  ``"_duration_ns"`` is added per kernel."""
    db = sqlite3.connect(path)
    out = {}
    q = ("select name, counter_name, count(*), avg(v), avg(d) from (select name, counter_name, dispatch_id, "
         "sum(counter_value) as v, max(duration) as d from pmc_events group by name, counter_name, dispatch_id) "
         "group by name, counter_name")
    for name, ctr, n, mean, dur in db.execute(q):
        k = out.setdefault(norm(name), {})
        k[ctr] = (n, mean)
        k["_duration_ns"] = (n, dur)
    return out


def traffic(fetch_db, write_db, out, note=""):
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    kernels = {}
    for k in sorted(set(f) | set(w)):
        n, fk = f.get(k, {}).get("FETCH_SIZE", (0, 0.0))
        n = n or w.get(k, {}).get("WRITE_SIZE", (0, 0.0))[0]
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
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c["_duration_ns"][1] > 0:
            # matrix-core utilisation: busy cycles summed over the chip / (kernel duration x SIMDs x clock).  The
            # clock is the nominal 2.4 GHz (DVFS lowers it under load: the true utilisation is a little higher)
            row["mfma_util_at_2.4GHz"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (c["_duration_ns"][1] * 2.4 * 1024), 4)
            if c.get("SQ_INSTS_VALU", (0, 0))[1] > 0 and "SQ_INSTS_VALU_MFMA_MOPS_F32" in c:
                row["mfma_ops_per_valu_inst"] = round(c["SQ_INSTS_VALU_MFMA_MOPS_F32"][1] / c["SQ_INSTS_VALU"][1], 4)
        res[name] = row
    json.dump({"command": "rocprofv3 --kernel-trace --pmc <counters> (tools/pmc.sh) -- python3 bench.py ... " + note,
               "unit": "counter value per launch (summed over the hardware instances of a dispatch), averaged over the "
                       "launches of a kernel; SQ_VALU_MFMA_BUSY_CYCLES = 8 cycles per v_mfma_f32_16x16x4_f32",
               "kernels": res}, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:6])
    else:
        counters(*sys.argv[2:5])
