#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as a per-kernel table.

    python tools/rocpd_summary.py gpurun_out/prof/xyz_results.db [--per-grid] > profiles/xyz.md

Equivalent of the `--stats` kernel summary (calls, total, average, min, max, share) for
runs where rocprofv3 wrote the default rocpd database.
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main():
    path = sys.argv[1]
    per_grid = "--per-grid" in sys.argv
    db = sqlite3.connect(path)
    rows = db.execute("select name, duration, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size from kernels").fetchall()
    agg = {}
    for name, dur, gx, wx, vg, ag, lds in rows:
        key = (short(name), gx // max(wx, 1)) if per_grid else (short(name),)
        a = agg.setdefault(key, [0, 0, 1 << 62, 0, vg, ag, lds])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    hdr = "| kernel |" + (" workgroups |" if per_grid else "") + " calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B |"
    print(hdr)
    print("|" + "---|" * (hdr.count("|") - 1))
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        cells = [key[0]] + ([str(key[1])] if per_grid else [])
        cells += [str(a[0]), f"{a[1] / 1e6:.3f}", f"{a[1] / a[0] / 1e3:.1f}", f"{a[2] / 1e3:.1f}", f"{a[3] / 1e3:.1f}",
                  f"{100.0 * a[1] / total:.1f}", str(a[4]), str(a[5]), str(a[6])]
        print("| " + " | ".join(cells) + " |")
    print(f"\ntotal kernel time {total / 1e6:.3f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
