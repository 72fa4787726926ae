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
SOURCES = ("zett_hip.hip",)
HEADERS = tuple(sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))) + ("../../include/zett_hip.h",)      # every header of csrc/
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC")


def source_hash() -> str:
    """sha256 over the HIP sources the library is built from: ties a measurement (profiles/pmc_traffic.json) to the
    kernels it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for rel in sorted(SOURCES + tuple(f for f in HEADERS if not f.startswith(".."))):
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


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc(), *HIPCC_FLAGS, "-o", LIB_PATH + ".tmp", *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print("[zett_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
