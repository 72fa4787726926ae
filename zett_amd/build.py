"""Build libzett_hip.so for gfx950 with hipcc, in-tree.

    python -m zett_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so sits next to the sources
(zett_amd/csrc/libzett_hip.so) so it travels with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libzett_hip.so")
# One translation unit per kernel family and operand type: hipcc compiles them in parallel, and an edit to one tile kernel
# rebuilds only its own objects (csrc/gemm_launch.hip.h).
SOURCES = tuple(sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")))
HEADERS = tuple(sorted(f for f in os.listdir(CSRC) if f.endswith(".h") or f.endswith(".inc"))) + ("../../include/zett_hip.h",)      # every header of csrc/
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC")
OBJ_DIR = os.path.join(CSRC, "build")


TRAINING_ONLY = ("train_ops.hip",)      # primitives of zett_amd/autograd.py: not on the path bench.py measures


def source_hash() -> str:
    """sha256 over the HIP sources the FORWARD (what bench.py measures) is built from: ties a measurement
    (profiles/pmc_traffic.json) to the kernels it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for rel in sorted(tuple(f for f in SOURCES if f not in TRAINING_ONLY) + tuple(f for f in HEADERS if not f.startswith(".."))):
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libzett_hip.so cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    for rel in SOURCES + HEADERS:
        path = os.path.join(CSRC, rel)
        if os.path.exists(path) and os.path.getmtime(path) > built:
            return True
    return False


def _includes(path: str, seen=None) -> set:
    """Headers of csrc/ a source reaches through #include "..." (transitively)."""
    import re
    seen = set() if seen is None else seen
    with open(path) as f:
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', f.read(), re.M):
            full = os.path.normpath(os.path.join(os.path.dirname(path), inc))
            if os.path.exists(full) and full not in seen:
                seen.add(full)
                _includes(full, seen)
    return seen


def _object_stale(src: str, obj: str) -> bool:
    if not os.path.exists(obj):
        return True
    built = os.path.getmtime(obj)
    return any(os.path.getmtime(p) > built for p in [src, *_includes(src)])


def remarks_path(source_name: str) -> str:
    """where build() keeps hipcc's -Rpass-analysis=kernel-resource-usage output of one translation unit"""
    return os.path.join(OBJ_DIR, source_name[:-4] + ".remarks")


def fresh_remarks(source_name: str):
    """The resource-usage remarks of a translation unit as the last build wrote them, or None if its sources changed since."""
    src, path = os.path.join(CSRC, source_name), remarks_path(source_name)
    if not os.path.exists(path) or _object_stale(src, path):
        return None
    with open(path) as f:
        return f.read()


LAST_BUILD = {"hipcc_commands": 0}      # what the most recent build() did (printed by __graft_entry__.build)


def build(force: bool = False, verbose: bool = True, jobs: int = 0) -> str:
    LAST_BUILD["hipcc_commands"] = 0
    if not force and not is_stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ_DIR, s[:-4] + ".o")
        objs.append(obj)
        if force or _object_stale(src, obj):
            todo.append([hipcc, *HIPCC_FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[zett_amd.build]", " ".join(cmd), flush=True)
        if "-c" not in cmd:
            subprocess.run(cmd, check=True, cwd=CSRC)
            return
        # the compiler's per-kernel resource-usage remarks (registers, LDS, scratch) of this translation unit are kept beside its
        # object: tests/test_host_logic.py::test_no_product_kernel_uses_scratch reads them instead of compiling everything again
        out = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
        if out.returncode != 0:
            sys.stderr.write(out.stderr)
            raise subprocess.CalledProcessError(out.returncode, cmd)
        if verbose and ("warning:" in out.stderr or "error:" in out.stderr):          # (the remarks and their source-context lines stay in the file)
            import re
            other = [l for l in out.stderr.splitlines() if "remark:" not in l and l.strip() and not re.match(r"^\s*(\d+\s*)?\|", l)]
            sys.stderr.write("\n".join(other) + "\n")
        name = os.path.basename(cmd[cmd.index("-c") + 1])
        with open(remarks_path(name) + ".tmp", "w") as f:
            f.write(out.stderr)
        os.replace(remarks_path(name) + ".tmp", remarks_path(name))

    jobs = jobs or int(os.environ.get("ZETT_BUILD_JOBS", "0")) or min(len(todo) or 1, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
        list(pool.map(run, todo))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH + ".tmp", *objs])
    LAST_BUILD["hipcc_commands"] = len(todo) + 1
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
