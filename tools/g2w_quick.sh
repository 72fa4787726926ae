#!/bin/bash
# GPU box: a short gemm2w vs gemm4d comparison (plain 16-bit output).  gpurun -- 'bash tools/g2w_quick.sh'
B=$GRAFT_REPO_ROOT/tools/gemm_bench_lean
export NORMAL=1 BURST=10 ROUNDS=3 ONLY=${ONLY:-p4d,g2w}
EPI=${EPI:-0} timeout 300 $B 169283 2304 768 169283 768 1536 118979 6144 2048 9700 12288 4096 77450 4096 4096 77450 4096 8192 2>&1
