#!/bin/bash
# GPU box: only the headline's kernel-statistics pass, PMC passes and launch log of tools/profile_round.sh (into the same directory),
# for when the rest of a set is already there.   gpurun -- 'bash tools/profile_headline_only.sh r5e'
set -u
tag=${1:-rX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
bench="python $GRAFT_REPO_ROOT/bench.py --no-live-traffic --no-side-configs"
rm -rf $out/prof $out/pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o $tag -- $bench --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision > $out/prof_bench.json 2> $out/prof.err
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$out/pmc_$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o t -- $bench --steps 1 --warmup 1 --no-cpu-baseline --no-alt-precision > /dev/null 2> $d.err
done
ZETT_GEMM_LOG=1 $bench --steps 1 --warmup 1 --no-cpu-baseline --no-alt-precision 2> $out/gemm_launch_log.txt > /dev/null
find $out -name "*.db" -delete; find $out -name "*agent_info*" -delete
du -sh $out
