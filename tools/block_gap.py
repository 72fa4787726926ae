#!/usr/bin/env python
"""Where a forward's plan ran and how long the GPU idled between the row blocks of a sharded step, from a
`rocprofv3 --kernel-trace --output-format csv` trace of `bench.py --gpus 1 --force-gather --chunks N` (or of any multi-block run).

    python tools/block_gap.py <..._kernel_trace.csv>

A forward = the kernels from its `gather_src_kernel` (first launch behind the plan) to its last GEMM.  For every forward after the
first of a step: the gap between the previous forward's last kernel and this forward's first kernel, and whether this forward's
plan (`plan_rows_kernel`) started before the previous forward ended (zett_forward_prepare: it should)."""
import csv
import sys


def main(path):
    rows = [r for r in csv.DictReader(open(path)) if "zett" in r["Kernel_Name"]]      # (rocprofv3 leaves the half-precision instances mangled: _ZN4zett...)
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows))
    plan = [e for e in ev if "plan_rows_kernel" in e[2]]
    gath = [e for e in ev if "gather_src_kernel" in e[2] and "convert" not in e[2]]
    body = [e for e in ev if not any(k in e[2] for k in ("plan_", "scan", "exclusive_scan", "retok", "chars_to_bytes", "token_raw", "fill_i32"))]
    out = []
    for g in gath[1:]:
        prev = [e for e in body if e[1] <= g[0]]
        if not prev:
            continue
        last = max(prev, key=lambda e: e[1])
        gap_us = (g[0] - last[1]) / 1e3
        p = [e for e in plan if e[0] <= g[0]]
        ahead_us = (last[1] - p[-1][0]) / 1e3 if p else None      # > 0: the plan started that long BEFORE the previous forward's last kernel ended
        out.append((gap_us, ahead_us, p[-1][3] != g[3] if p else None))
    inside = [o for o in out if o[0] < 2000]      # (gaps of a whole host-side step boundary — warm-up / timing code — are not block boundaries)
    n_ahead = sum(1 for o in inside if o[1] is not None and o[1] > 0)
    gaps = sorted(o[0] for o in inside)
    print(f"forwards: {len(gath)}, boundaries inside steps: {len(inside)}; plan of the next block started before the previous forward's last kernel ended: {n_ahead}; "
          f"on another queue than the forward: {sum(1 for o in inside if o[2])}")
    if gaps:
        print(f"GPU idle between the previous forward's last kernel and the next forward's first kernel (us): min {gaps[0]:.1f}  median {gaps[len(gaps) // 2]:.1f}  max {gaps[-1]:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
