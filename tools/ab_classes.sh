#!/bin/bash
# GPU box: as tools/ab.sh, but prints the per-class GEMM table of each flag set (one run each).
for flags in "$@"; do
  python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-alt-precision --no-live-traffic --no-side-configs $flags 2>/dev/null | tail -1 | FLAGS="$flags" python -c '
import json, os, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print("== %s: %.2f ms/step (%.2f uninstr.), GEMM %.2f ms, frac %.4f" % (os.environ["FLAGS"], d["ms_per_step"], d["ms_per_step_uninstrumented"], r["gemm_ms_per_step"], r["frac"]))
for c in r["by_class"]:
    print("   %-100s %4.1f launches %7.3f ms  %6.0f TF" % (c["class"][:100], c["launches_per_step"], c["ms_per_step"], c["achieved"]))'
done
