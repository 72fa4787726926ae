#!/bin/bash
# GPU box: the product gemm4d kernel (16-bit-out epilogue) against a probe copy whose workgroups walk the tiles in a loop (one
# workgroup per CU, no prefetch of the next tile: tools/_ablate/src, built in the build container), alternating on one box.
#   gpurun --timeout 600 -- 'bash tools/persist_probe.sh'  -> gpurun_out/persist_probe.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/persist_probe.txt; : > $out
cd /tmp
for shape in "169283 1536 768" "110175 2304 768" "169283 768 1536" "118979 4096 2048" "118979 2048 4096" "77450 8192 4096" "9682 4096 4096"; do
  for rep in 1 2; do
    for b in gemm4d_ablate_L5 gemm4d_persist_L5; do
      echo -n "$b $shape: " >> $out
      $GRAFT_REPO_ROOT/tools/_ablate/$b $shape 2 0 2>&1 | tail -1 >> $out
    done
  done
done
cat $out
