#!/bin/bash
# Build container: gpurun_out/<tag>_<workload>/ (written by tools/profile_workload.sh on the GPU box) -> profiles/<tag>_<workload>_*.
#   bash tools/profile_workload_docs.sh r4a xlmr_gpt2 "one-line title"
set -eu
tag=$1; w=$2; title=${3:-}
src=gpurun_out/${tag}_$w; pre=profiles/${tag}_$w; H=$(git rev-parse --short HEAD)
cp $src/prof/${tag}_kernel_stats.csv ${pre}_kernel_stats.csv
cp $src/bench.json ${pre}_bench.json
grep "zett gemm" $src/gemm_launch_log.txt > ${pre}_gemm_launch_log.txt
{
  echo "# ${tag} / $w — $title (commit $H)"; echo
  echo "Command (MI355X box, \`tools/profile_workload.sh $tag $w\`): \`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --workload $w --no-live-traffic --no-cpu-baseline --no-alt-precision --steps 3 --warmup 1\`"
  echo "(8 forward passes in the trace: 1 warm-up + 3 timed with per-launch HIP events, then 1 + 3 without them; f16.  The \`__amd_rocclr_copyBuffer\`, \`at::native\` and \`convert_f32_to_lo\` rows are the untimed set-up.)  Source: \`${tag}_${w}_kernel_stats.csv\` as written by rocprofv3."; echo
  echo "bench line of the same (profiled) run:"; echo; echo '```'; cat $src/prof_bench.json; echo '```'; echo
  echo "un-profiled run on the same box: \`${pre}_bench.json\`."; echo
  python tools/kernel_stats_md.py ${pre}_kernel_stats.csv; echo
  python - "${pre}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
g = [r for r in rows if "gemm" in r["Name"]]
# (rocprofv3 writes some names demangled — "zett::retok_tokens_kernel(...)" — and some mangled — "_ZN4zett16gemm4d_tn_kernel...")
z = [r for r in rows if ("zett::" in r["Name"] or "_ZN4zett" in r["Name"]) and "convert_f32" not in r["Name"] and "fold_weight" not in r["Name"]]
assert all(r in z for r in g), "a GEMM kernel the path filter does not see"
tot = sum(int(r["TotalDurationNs"]) for r in g); calls = sum(int(r["Calls"]) for r in g); allz = sum(int(r["TotalDurationNs"]) for r in z)
print(f"GEMM kernels (all tile variants): {calls} launches, {tot / 1e6:.1f} ms in 8 forwards = {tot / 8e6:.2f} ms per forward; every kernel of the path "
      f"(zett::*): {allz / 8e6:.2f} ms per forward, of which {100 * (allz - tot) / allz:.1f} % is not a GEMM.")
PY
  echo; echo "Per-launch log of one forward (\`ZETT_GEMM_LOG=1\`, HIP events on the launch stream): \`${pre}_gemm_launch_log.txt\`."
} > ${pre}_kernel_stats.md
{
  echo "# ${tag} / $w — PMC passes on bench.py --workload $w (commit $H)"; echo
  echo "Three separate passes (\`--pmc\` only with \`--kernel-trace\`, as the MI355X guide prescribes), each:"
  echo "\`rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv -- python bench.py --workload $w --steps 1 --warmup 1 ...\`"
  echo "with COUNTERS = \`FETCH_SIZE\` | \`WRITE_SIZE\` | \`SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE\` (f16; four forwards per pass)."
  echo "Summarised by \`tools/pmc_summary.py\` (FETCH_SIZE doubled per the gfx950 correction; GRBM_GUI_ACTIVE is summed over the 8 XCDs)."; echo
  python tools/pmc_summary.py $src/pmc_FETCH_SIZE $src/pmc_WRITE_SIZE $src/pmc_SQ_VALU_MFMA_BUSY_CYCLES
} > ${pre}_pmc.md
echo "wrote ${pre}_*"
