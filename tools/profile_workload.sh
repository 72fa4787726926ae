#!/bin/bash
# GPU box: rocprofv3 kernel statistics + the three PMC passes + the per-launch GEMM log for ONE named workload
# (tools/profile_round.sh does this for the default line only).
#   gpurun --timeout 900 -- 'bash tools/profile_workload.sh r4a xlmr_gpt2 [extra bench.py flags]'
set -u
tag=${1:-rX}; w=${2:-xlmr_gpt2}; shift 2
extra="$*"
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_$w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
bench="python $GRAFT_REPO_ROOT/bench.py --workload $w --no-live-traffic --no-cpu-baseline --no-alt-precision --no-side-configs $extra"
$bench 2>/dev/null | tail -1 > $out/bench.json
cp $GRAFT_REPO_ROOT/bench_side.json $out/bench_side.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o $tag -- $bench --steps 3 --warmup 1 > $out/prof_bench.json 2> $out/prof.err
cp $GRAFT_REPO_ROOT/bench_side.json $out/prof_bench_side.json
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$out/pmc_$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o t -- $bench --steps 1 --warmup 1 > /dev/null 2> $d.err
done
ZETT_GEMM_LOG=1 $bench --steps 1 --warmup 1 2> $out/gemm_launch_log.txt > /dev/null
find $out -name "*.db" -delete; find $out -name "*agent_info*" -delete
du -sh $out
