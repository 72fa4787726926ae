set -u
out=$GRAFT_REPO_ROOT/gpurun_out/s2b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --rows 4096 --no-cpu-baseline --no-alt-precision --no-live-traffic --no-side-configs"
$B 2>/dev/null | tail -1 > $out/rows4096.json
$B --partition affinity 2>/dev/null | tail -1 > $out/rows4096_aff.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_cont -o t -- $B --steps 4 --warmup 2 > $out/p_cont.json 2> $out/p_cont.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p_aff -o t -- $B --partition affinity --steps 4 --warmup 2 > $out/p_aff.json 2> $out/p_aff.err
find $out -name "*.db" -delete; find $out -name "*agent_info*" -delete
cd $GRAFT_REPO_ROOT
timeout 400 python tools/retok_fuzz.py --leg gpu --seeds 0 100000 --budget-s 300 > $out/retok_fuzz_gpu.json 2> $out/retok_fuzz_gpu.err
tail -2 $out/retok_fuzz_gpu.json; cat $out/rows4096.json | cut -c1-600; du -sh $out
