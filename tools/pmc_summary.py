#!/usr/bin/env python
"""Per-kernel summary of rocprofv3 --pmc counter_collection CSVs (one directory per counter pass).

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_MFMA

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md §HBM), so the read figure printed here is FETCH_SIZE * 2 * 1024 bytes; WRITE_SIZE
is used as reported.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs).
"""
import collections
import csv
import glob
import re
import os
import sys


import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from kname import short as _short


def short(name):
    return _short(name, 0)[:60]


def main(dirs):
    data = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for path in glob.glob(d + "/*counter_collection.csv"):
            for r in csv.DictReader(open(path)):
                k = short(r["Kernel_Name"])
                data[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                data[k]["_dur_" + r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("| kernel | launches | avg us | HBM read GB/launch (FETCH_SIZE x2) | HBM write GB/launch | read+write GB/s | MFMA util | clock GHz |")
    print("|---|---|---|---|---|---|---|---|")
    rows = []
    for k, v in data.items():
        if "zett::" not in k:
            continue
        n = max(len(x) for x in v.values())
        def avg(c):
            return sum(v[c]) / len(v[c]) if c in v and v[c] else None
        fetch, write = avg("FETCH_SIZE"), avg("WRITE_SIZE")
        dur = avg("_dur_FETCH_SIZE") or avg("_dur_WRITE_SIZE") or avg("_dur_GRBM_GUI_ACTIVE")
        rd = None if fetch is None else fetch * 2 * 1024 / 1e9
        wr = None if write is None else write * 1024 / 1e9
        util = clk = None
        if avg("SQ_VALU_MFMA_BUSY_CYCLES") is not None and avg("GRBM_GUI_ACTIVE"):
            cyc = avg("GRBM_GUI_ACTIVE") / 8
            util = avg("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024)
            clk = cyc / avg("_dur_GRBM_GUI_ACTIVE")
        bw = None if rd is None or wr is None or not dur else (rd + wr) / (dur * 1e-9)
        rows.append((-(dur or 0) * n, k, n, dur, rd, wr, bw, util, clk))
    if "--json" in sys.argv:     # launch-weighted HBM bytes per GEMM launch, for bench.py's roofline.traffic
        import json
        tot_b = tot_n = 0.0
        for _, k, n, dur, rd, wr, bw, util, clk in rows:
            if "gemm" in k and rd is not None and wr is not None:
                tot_b += (rd + wr) * 1e9 * n
                tot_n += n
        out = {"kernel": "zett::gemm*_tn_kernel (all instances, launch-weighted)", "hbm_bytes_per_launch": tot_b / max(tot_n, 1),
               "launches": int(tot_n), "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950)",
               "workload": "bench.py default (mistral_gpt2_32k)", "precision": os.environ.get("ZETT_PMC_PRECISION", "f16")}
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from zett_amd.build import source_hash
        out["source_hash"] = source_hash()
        out["tag"] = os.environ.get("ZETT_PMC_TAG")                  # profile set (profiles/<tag>_*) the passes belong to
        try:
            import subprocess
            out["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout.strip() or None
        except Exception:
            out["commit"] = None
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    f = lambda x, fmt: "" if x is None else fmt % x
    for _, k, n, dur, rd, wr, bw, util, clk in sorted(rows):
        print(f"| {k} | {n} | {f(dur and dur / 1e3, '%.1f')} | {f(rd, '%.3f')} | {f(wr, '%.3f')} | {f(bw, '%.0f')} | {f(util, '%.3f')} | {f(clk, '%.2f')} |")


if __name__ == "__main__":
    main([a for i, a in enumerate(sys.argv[1:]) if a != "--json" and (i == 0 or sys.argv[i] != "--json")])
