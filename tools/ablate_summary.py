#!/usr/bin/env python
"""Build container: gpurun_out/<tag>_ablation/ (tools/ablate.sh) -> the table of profiles/<tag>_ablation.md."""
import csv
import glob
import json
import os
import re
import sys

src = sys.argv[1]
NAMES = {"L1": "(i) the MFMAs of the K loop alone", "L2": "(ii) + fragment reads (ds_read_b128, 32 per wave and K step)", "L3": "(iii) + LDS-DMA requests (16 per wave and K step)",
         "L4": "(iv) + waits and barriers = the whole K loop", "L5": "(v) + epilogue (16-bit output) = the product kernel", "L4_zero": "(iv) on all-zero operands", "L5_zero": "(v) on all-zero operands"}
for _k in ("L1", "L2", "L3", "L4", "L4_zero"):
    NAMES["M32_" + _k] = NAMES[_k] + " — on v_mfma_f32_32x32x16 (64 MFMAs per K step)"


def smi(path):
    rows = []
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        try:
            c = json.loads(line).get("card0", {})
        except Exception:
            continue

        def num(pat):
            for k, v in c.items():
                if re.search(pat, k, re.I):
                    m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
                    if m:
                        return float(m.group(0))
            return None
        rows.append((num(r"power"), num(r"sclk")))
    rows = [r for r in rows if r[0]]
    if not rows:
        return None, None, 0
    top = max(r[0] for r in rows)
    busy = sorted(r for r in rows if r[0] > 0.6 * top)
    med = lambda v: sorted(v)[len(v) // 2]
    return med([r[0] for r in busy]), med([r[1] for r in busy if r[1]]), len(busy)


def pmc(dirname):
    tot = {}
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "gemm4d" in r.get("Kernel_Name", ""):
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and tot.get("GRBM_GUI_ACTIVE"):
        # MFMA_BUSY is summed over the 1024 SIMDs (4 per CU x 256), GRBM_GUI_ACTIVE over the 8 XCDs
        return tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (tot["GRBM_GUI_ACTIVE"] / 8.0)
    return None


print("| what runs | TFLOP/s | of the 2.5 PFLOP/s roof | socket power (W, median under load) | shader clock (MHz, median) | MFMA peak at that clock (TFLOP/s) | of THAT peak | SQ_VALU_MFMA_BUSY (PMC pass) |")
print("|---|---|---|---|---|---|---|---|")
for key in ("L1", "M32_L1", "L2", "M32_L2", "L3", "M32_L3", "L4", "M32_L4", "L5", "L4_zero", "M32_L4_zero", "L5_zero"):
    p = os.path.join(src, key + ".json")
    if not os.path.exists(p) or not open(p).read().strip():
        continue
    d = json.loads(open(p).read().strip().splitlines()[-1])
    w, clk, n = smi(os.path.join(src, key + ".smi.txt"))
    busy = pmc(os.path.join(src, "pmc_" + key)) if "zero" not in key else None
    peak = 2500.0 * clk / 2400.0 if clk else None
    print(f"| {NAMES[key]} | {d['tflops']:.0f} | {d['frac_of_2500']:.3f} | {w if w else '-'} ({n} samples) | {clk if clk else '-'} | {peak:.0f} | {d['tflops'] / peak:.3f} | {('%.3f' % busy) if busy else '-'} |"
          if peak else f"| {NAMES[key]} | {d['tflops']:.0f} | {d['frac_of_2500']:.3f} | - | - | - | - | - |")
