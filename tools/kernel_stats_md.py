#!/usr/bin/env python
"""rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv -> markdown table.

    python tools/kernel_stats_md.py gpurun_out/prof_r1d/*/r1d_kernel_stats.csv > table.md
"""
import csv
import re
import sys


import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from kname import short as _short


def short(name):
    return _short(name, 40)


def main(path):
    rows = list(csv.DictReader(open(path)))
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        if float(r["Percentage"]) < 0.005:
            continue
        print(f"| {short(r['Name'])} | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
              f"{int(r['MinNs']) / 1e3:.1f} | {int(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
