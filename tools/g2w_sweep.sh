#!/bin/bash
# GPU box: gemm2w (128x256, two workgroups per CU) against gemm4d / gemm8r on the launch shapes of the narrow workloads,
# of a vocabulary shard and of the headline step.  gpurun -- 'bash tools/g2w_sweep.sh r3b'
tag=${1:-rX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
B=$GRAFT_REPO_ROOT/tools/gemm_bench_lean
export NORMAL=1 BURST=10 ROUNDS=3 ONLY=g8r,p4d,g2w,g2wg
XLMR="169283 2304 768 169283 768 768 169283 1536 768 169283 768 1536"
TINY="118979 6144 2048 118979 2048 2048 118979 4096 2048 118979 2048 4096"
SHARD="9700 12288 4096 9700 4096 4096 9700 8192 4096 9700 4096 8192 4096 4096 4096"
HEAD="77450 4096 4096 77450 8192 4096 77450 4096 8192"
for epi in 0 5 1; do
  EPI=$epi timeout 600 $B $XLMR $TINY $SHARD $HEAD 2>&1 | tee -a $out/g2w_sweep.txt
done
