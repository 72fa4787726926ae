set -u
out=$GRAFT_REPO_ROOT/gpurun_out/s2i; mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-alt-precision --no-live-traffic > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s2i/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], len(json.dumps(d)))
for c in d.get('configs',[]): print({k:v for k,v in c.items() if k!='cpu_baseline'})
PY
ZETT_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --workload xlmr_gpt2 --rows 20001 --no-cpu-baseline --table-exchange 2> $out/two.err | tail -1 | cut -c1-1500; tail -3 $out/two.err
ZETT_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --workload xlmr_gpt2 --rows 20001 --no-cpu-baseline --table-exchange --partition affinity --gather-mode fanout 2> $out/two2.err | tail -1 | cut -c1-600; tail -3 $out/two2.err
