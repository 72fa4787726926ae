#!/usr/bin/env python
"""Build container: the probe binaries of tools/persist_probe.sh (r6).  Copies zett_amd/csrc/gemm4d.hip.h to tools/_ablate/src/
with three edits — the kernel body inside `for (tile = blockIdx.x; tile < tiles; tile += gridDim.x)`, a barrier at the end of a
tile (LDS is reused by the next prologue), an opaque thread id per tile (so that lane-derived values are recomputed instead of being
hoisted across the tile loop and spilled), a grid of at most 256 workgroups — and compiles tools/gemm4d_ablate.hip against the
product header (gemm4d_ablate_L5) and against the copy (gemm4d_persist_L5).  The product sources are not touched.

    python tools/make_persist_probe.py        # then: gpurun -- 'bash tools/persist_probe.sh'
"""
import os
import shutil
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "zett_amd", "csrc")
OUT = os.path.join(REPO, "tools", "_ablate")
SRC = os.path.join(OUT, "src")


def patched_header() -> str:
    s = open(os.path.join(CSRC, "gemm4d.hip.h")).read()

    def sub(old, new):
        nonlocal s
        assert s.count(old) == 1, old
        s = s.replace(old, new)

    sub("    int wg = blockIdx.x;\n    {\n        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;",
        "    for (int wg_it = blockIdx.x; wg_it < nwg; wg_it += gridDim.x) {          // PROBE: one workgroup per CU walks the tiles\n"
        "    int wg = wg_it;\n    {\n        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;")
    sub("        range_report(e.range_flag, bad, W16 ? ZETT_RANGE_BIT_ACTIVATION : ZETT_RANGE_BIT_OUTPUT);\n    }\n}\n\n// Which epilogue a launch gets.",
        "        range_report(e.range_flag, bad, W16 ? ZETT_RANGE_BIT_ACTIVATION : ZETT_RANGE_BIT_OUTPUT);\n    }\n"
        "    __builtin_amdgcn_s_waitcnt(G4R_WAIT_LGKM0);          // PROBE: every wave is done with the staged tile before the next prologue overwrites LDS\n"
        "    __builtin_amdgcn_s_barrier();\n    }\n}\n\n// Which epilogue a launch gets.")
    sub("hipLaunchKernelGGL((gemm4d_tn_kernel<T, ACT, RES, EPI, HALF>), dim3(tiles_m * tiles_n), dim3(256), lds, stream, g);",
        "hipLaunchKernelGGL((gemm4d_tn_kernel<T, ACT, RES, EPI, HALF>), dim3(tiles_m * tiles_n < 256 ? tiles_m * tiles_n : 256), dim3(256), lds, stream, g);")
    sub("    const int tid = threadIdx.x;\n    const int lane = tid & 63;",
        "    int tid = threadIdx.x;\n    asm volatile(\"\" : \"+v\"(tid));          // PROBE: lane-derived values recomputed per tile\n    const int lane = tid & 63;")
    return s


def main():
    os.makedirs(SRC, exist_ok=True)
    open(os.path.join(SRC, "gemm4d.hip.h"), "w").write(patched_header())
    shutil.copy(os.path.join(REPO, "tools", "gemm4d_ablate.hip"), os.path.join(SRC, "gemm4d_ablate_persist.hip"))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-DG4D_ABLATE=5"]
    subprocess.run([hipcc, *flags, "-I", CSRC, os.path.join(REPO, "tools", "gemm4d_ablate.hip"), "-o", os.path.join(OUT, "gemm4d_ablate_L5")], check=True)
    out = subprocess.run([hipcc, *flags, "-I", SRC, "-I", CSRC, os.path.join(SRC, "gemm4d_ablate_persist.hip"), "-o", os.path.join(OUT, "gemm4d_persist_L5"),
                          "-Rpass-analysis=kernel-resource-usage"], check=True, capture_output=True, text=True)
    for line in out.stderr.splitlines():
        if any(k in line for k in ("Function Name", " VGPRs:", "AGPRs:", "ScratchSize")):
            print(line.split("remark: ")[-1].split(" [-R")[0])


if __name__ == "__main__":
    main()
