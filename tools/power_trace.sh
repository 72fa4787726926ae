#!/bin/bash
# GPU box: socket power and shader clock under the benchmark step, sampled with rocm-smi while bench.py runs many steps.
#   gpurun -- 'bash tools/power_trace.sh r4 [workload]'   -> gpurun_out/<tag>_power_<workload>.txt (+ .json summary)
set -u
tag=${1:-rX}; w=${2:-mistral_gpt2_32k}
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_power_$w
cd /tmp
python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 150 --warmup 3 --no-cpu-baseline --no-alt-precision --no-live-traffic --no-side-configs > $out.bench.json 2> /dev/null &
pid=$!
: > $out.txt
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi -P -c -u --json 2>/dev/null | tr -d '\n' >> $out.txt; echo >> $out.txt
  sleep 0.05
done
python - $out.txt $out.json <<'PY'
import json, sys, re
rows = []
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    c = d.get("card0", {})
    def num(pattern):
        for k, v in c.items():
            if re.search(pattern, k, re.I):
                m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
                if m:
                    return float(m.group(0))
        return None
    rows.append({"power_w": num(r"power"), "sclk_mhz": num(r"sclk clock speed|sclk"), "use_pct": num(r"GPU use")})
busy = [r for r in rows if r["power_w"] and r["power_w"] > 0.6 * max(x["power_w"] or 0 for x in rows)]
def stat(key, rs):
    v = sorted(x[key] for x in rs if x[key] is not None)
    return None if not v else {"min": v[0], "median": v[len(v) // 2], "max": v[-1], "mean": sum(v) / len(v), "samples": len(v)}
json.dump({"samples": len(rows), "under_load": {"power_w": stat("power_w", busy), "sclk_mhz": stat("sclk_mhz", busy)},
           "keys_seen": sorted(set().union(*[set(json.loads(l).get("card0", {}).keys()) for l in open(sys.argv[1]) if l.strip().startswith("{")][:1] or [set()]))},
          open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
PY
