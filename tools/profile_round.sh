#!/bin/bash
# Runs on the MI355X box (through gpurun): the bench lines, the rocprofv3 kernel-trace/stats pass
# and the three PMC passes whose summaries are committed under profiles/ (named per round).
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r1d'
set -u
tag=${1:-rX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
bench="python $GRAFT_REPO_ROOT/bench.py --no-live-traffic --no-side-configs"      # (the PMC passes below are the profile set; the default line measures its own traffic and carries the side lines)
python $GRAFT_REPO_ROOT/bench.py 2>/dev/null | tail -1 > $out/bench_default.json
cp $GRAFT_REPO_ROOT/bench_side.json $out/bench_default_side.json      # (r6: stdout is the compact line; the full objects — per-class tables of every side config — are in bench_side.json)
for w in xlmr_gpt2 tinyllama_neox mistral_neox llama3_256k; do $bench --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_$w.json; cp $GRAFT_REPO_ROOT/bench_side.json $out/bench_${w}_side.json; done
$bench --precision f32 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_default_f32.json
$bench --precision bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_default_bf16.json
for r in 16384 8192 4096; do $bench --rows $r --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_rows_$r.json; cp $GRAFT_REPO_ROOT/bench_side.json $out/bench_rows_${r}_side.json; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o $tag -- $bench --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precision > $out/prof_bench.json 2> $out/prof.err
cp $GRAFT_REPO_ROOT/bench_side.json $out/prof_bench_side.json
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$out/pmc_$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o t -- $bench --steps 1 --warmup 1 --no-cpu-baseline --no-alt-precision > /dev/null 2> $d.err
done
ZETT_GEMM_LOG=1 $bench --steps 1 --warmup 1 --no-cpu-baseline --no-alt-precision 2> $out/gemm_launch_log.txt > /dev/null
find $out -name "*.db" -delete; find $out -name "*agent_info*" -delete
du -sh $out
# the narrow workloads get their own kernel tables and counter passes (tools/profile_workload.sh), and the headline a power / clock trace
for w in xlmr_gpt2 tinyllama_neox; do bash $GRAFT_REPO_ROOT/tools/profile_workload.sh $tag $w > $out/workload_$w.log 2>&1; done
bash $GRAFT_REPO_ROOT/tools/power_trace.sh $tag mistral_gpt2_32k > $out/power.log 2>&1
bash $GRAFT_REPO_ROOT/tools/power_trace.sh $tag xlmr_gpt2 >> $out/power.log 2>&1
