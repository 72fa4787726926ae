#!/bin/bash
# GPU box: VERDICT r4 item 5a — the product K loop of gemm4d piece by piece under RANDOM operands, with socket power, shader clock
# and the MFMA-busy counter beside the TFLOP/s (tools/gemm4d_ablate.hip; binaries prebuilt in tools/_ablate/ by the build container).
#   gpurun --timeout 900 -- 'bash tools/ablate.sh r5'   -> gpurun_out/<tag>_ablation/{L*.json, L*.smi.txt, pmc_L*/}
set -u
tag=${1:-rX}; M=${2:-77450}; N=${3:-8192}; K=${4:-4096}
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_ablation
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {   # name binary zero
  $GRAFT_REPO_ROOT/tools/_ablate/$2 $M $N $K 6 $3 > $out/$1.json 2> $out/$1.err &
  pid=$!
  : > $out/$1.smi.txt
  while kill -0 $pid 2>/dev/null; do
    /opt/rocm/bin/rocm-smi -P -c -u --json 2>/dev/null | tr -d '\n' >> $out/$1.smi.txt; echo >> $out/$1.smi.txt
    sleep 0.05
  done
  cat $out/$1.json
}
# (r6: the same levels with the K loop on v_mfma_f32_32x32x16 — -DG4D_MFMA32, levels 1-4 — run right behind their 16x16x32 twins: same box, same minute)
for L in 1 2 3 4 5; do
  run L$L gemm4d_ablate_L$L 0
  [ -x $GRAFT_REPO_ROOT/tools/_ablate/gemm4d_ablate32_L$L ] && run M32_L$L gemm4d_ablate32_L$L 0
done
run L4_zero gemm4d_ablate_L4 1
[ -x $GRAFT_REPO_ROOT/tools/_ablate/gemm4d_ablate32_L4 ] && run M32_L4_zero gemm4d_ablate32_L4 1
run L5_zero gemm4d_ablate_L5 1
for L in 1 2 3 4 5; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_L$L -o t -- $GRAFT_REPO_ROOT/tools/_ablate/gemm4d_ablate_L$L $M $N $K 0.4 0 > /dev/null 2> $out/pmc_L$L.err
  [ -x $GRAFT_REPO_ROOT/tools/_ablate/gemm4d_ablate32_L$L ] && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_M32_L$L -o t -- $GRAFT_REPO_ROOT/tools/_ablate/gemm4d_ablate32_L$L $M $N $K 0.4 0 > /dev/null 2> $out/pmc_M32_L$L.err
done
find $out -name "*.db" -delete; find $out -name "*agent_info*" -delete
du -sh $out
