#!/usr/bin/env python
"""One step of `bench.py` as a timeline, from a `rocprofv3 --kernel-trace --output-format csv` trace: every kernel of the LAST
step in launch order with its start offset, duration and the GPU idle time in front of it, then the totals by kernel family.

    python tools/step_timeline.py <..._kernel_trace.csv> [--steps-back 0] [--all]

A step starts at a `chars_to_bytes_kernel` (the retokenizer's first launch) — or, with `--no-retokenize` traces, at
`plan_rows_kernel` — and ends in front of the next one.  Made for the question "where do a 4 096-row shard's fixed costs go":
the `--stats` table gives sums per kernel, not the gaps between launches."""
import argparse
import csv
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
try:
    from kname import short as _short          # tools/kname.py demangles the half-precision template instances
except Exception:          # pragma: no cover
    _short = None


def short(name):
    if _short is not None:
        try:
            name = _short(name)
        except Exception:
            pass
    name = re.sub(r"^void ", "", name)
    name = name.replace("zett::", "")
    return name[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps-back", type=int, default=0, help="0 = the last step of the trace, 1 = the one before, ...")
    ap.add_argument("--all", action="store_true", help="print every kernel, not only those >= 3 us or behind a gap >= 2 us")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    starts = [i for i, e in enumerate(ev) if "chars_to_bytes_kernel" in e[2]]
    if not starts:
        starts = [i for i, e in enumerate(ev) if "plan_rows_kernel" in e[2]]
    if not starts:
        sys.exit("no step start (chars_to_bytes_kernel / plan_rows_kernel) in the trace")
    k = len(starts) - 1 - a.steps_back
    lo, hi = starts[k], (starts[k + 1] if k + 1 < len(starts) else len(ev))
    step = ev[lo:hi]
    t0 = step[0][0]
    fam = {}
    busy = 0
    prev_end = t0
    print(f"step {k} of {len(starts)}: {len(step)} kernels, {(max(e[1] for e in step) - t0) / 1e3:.1f} us from first start to last end")
    print(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  kernel")
    for s, e, n in step:
        gap = (s - prev_end) / 1e3
        d = (e - s) / 1e3
        nm = short(n)
        f = re.sub(r"<.*", "", nm)
        c = fam.setdefault(f, [0, 0.0, 0.0])
        c[0] += 1
        c[1] += d
        c[2] += max(gap, 0.0)
        busy += e - s
        if a.all or d >= 3.0 or gap >= 2.0:
            print(f"{(s - t0) / 1e3:9.1f} {d:8.1f} {gap:7.1f}  {nm}")
        prev_end = max(prev_end, e)
    span = (prev_end - t0) / 1e3
    print(f"\nkernel time {busy / 1e3:.1f} us of {span:.1f} us (idle between launches {span - busy / 1e3:.1f} us)")
    print(f"{'family':40} {'n':>4} {'us':>9} {'gap in front us':>16}")
    for f, (n, d, g) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{f:40} {n:4d} {d:9.1f} {g:16.1f}")


if __name__ == "__main__":
    main()
