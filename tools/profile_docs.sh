#!/bin/bash
# Turns gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the committed profiles/<tag>_* files.
#   bash tools/profile_docs.sh r1e "one-line title"
set -eu
tag=$1; title=${2:-}
src=gpurun_out/$tag; H=$(git rev-parse --short HEAD)
cp $src/prof/${tag}_kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp $src/bench_default.json profiles/${tag}_bench.json
for w in xlmr_gpt2 tinyllama_neox mistral_neox llama3_256k; do cp $src/bench_$w.json profiles/${tag}_bench_$w.json; done
cp $src/bench_default_f32.json profiles/${tag}_bench_mistral_gpt2_32k_f32.json
cp $src/bench_default_bf16.json profiles/${tag}_bench_mistral_gpt2_32k_bf16.json
for r in 16384 8192 4096; do cp $src/bench_rows_$r.json profiles/${tag}_bench_rows_$r.json; done
grep "zett gemm" $src/gemm_launch_log.txt | tail -39 > profiles/${tag}_gemm_launch_log.txt
{
  echo "# $tag — $title (commit $H)"; echo
  echo "Command (MI355X box, \`tools/profile_round.sh $tag\`): \`rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -o $tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline\`"
  echo "(8 forward passes in the trace: 1 warm-up + 3 timed with per-launch HIP events, then bench.py's 1 + 3 passes without them — the \`ms_per_step_uninstrumented\` figure; workload mistral_gpt2_32k, f16 = the default precision.  The trace covers the WHOLE process: the \`__amd_rocclr_copyBuffer\`, \`at::native\` and \`convert_f32_to_lo\` rows are the untimed set-up — random weights generated with torch on the GPU, ~100 weight uploads and conversions — not part of a step, which launches no torch kernel.)  Source: \`${tag}_kernel_stats.csv\` as written by rocprofv3."; echo
  echo "bench line of the same (profiled) run:"; echo; echo '```'; cat $src/prof_bench.json; echo '```'; echo
  echo "un-profiled default run on the same box (\`python bench.py\`, with the CPU baseline): \`profiles/${tag}_bench.json\`."; echo
  python tools/kernel_stats_md.py profiles/${tag}_kernel_stats.csv; echo
  python - "$tag" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f"profiles/{sys.argv[1]}_kernel_stats.csv")))
g = [r for r in rows if "gemm" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in g); calls = sum(int(r["Calls"]) for r in g)
print(f"GEMM kernels (all tile variants): {calls} launches, {tot / 1e6:.1f} ms in 8 forwards = {tot / 8e6:.2f} ms per forward "
      f"(bench.py's HIP-event figure for the same run: gemm_ms_per_step above), {sum(float(r['Percentage']) for r in g):.1f} % of the GPU time.")
PY
  echo; echo "Per-launch log of one forward (\`ZETT_GEMM_LOG=1\`, HIP events on the launch stream): \`profiles/${tag}_gemm_launch_log.txt\`."
} > profiles/${tag}_kernel_stats.md
{
  echo "# $tag — PMC passes on bench.py (commit $H)"; echo
  echo "Three separate passes (\`--pmc\` only with \`--kernel-trace\`, as the MI355X guide prescribes), each:"
  echo "\`rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv -d gpurun_out/$tag/pmc_X -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline\`"
  echo "with COUNTERS = \`FETCH_SIZE\` | \`WRITE_SIZE\` | \`SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE\` (workload mistral_gpt2_32k, f16; four forwards per pass: two with bench.py's per-launch HIP events, two without)."
  echo "Summarised by \`tools/pmc_summary.py\` (FETCH_SIZE doubled per the gfx950 correction; GRBM_GUI_ACTIVE is summed over the 8 XCDs)."; echo
  python tools/pmc_summary.py $src/pmc_FETCH_SIZE $src/pmc_WRITE_SIZE $src/pmc_SQ_VALU_MFMA_BUSY_CYCLES
} > profiles/${tag}_pmc.md
ZETT_PMC_TAG=$tag python tools/pmc_summary.py $src/pmc_FETCH_SIZE $src/pmc_WRITE_SIZE $src/pmc_SQ_VALU_MFMA_BUSY_CYCLES --json profiles/pmc_traffic.json > /dev/null
echo "wrote profiles/${tag}_*"
