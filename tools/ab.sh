#!/bin/bash
# GPU box: same-box A/B of bench.py flag sets.  gpurun -- 'bash tools/ab.sh "--ln-fold 1" "--ln-fold 2"'
# Each flag set runs twice, interleaved; prints ms_per_step (instrumented / uninstrumented), GEMM fraction, GEMM ms.
for rep in 1 2; do
  for flags in "$@"; do
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-alt-precision --no-live-traffic --no-side-configs $flags 2>/dev/null | tail -1 | FLAGS="$flags" python -c '
import json, os, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print("%-40s %7.2f %7.2f  frac %.4f  gemm %.2f ms" % (os.environ["FLAGS"], d["ms_per_step"], d["ms_per_step_uninstrumented"], r["frac"], r["gemm_ms_per_step"]))'
  done
done
