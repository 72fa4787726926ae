"""Readable kernel names for the rocprofv3 summaries: rocprofv3 leaves template kernels whose arguments involve _Float16
mangled (its demangler does not know the DF16_ builtin-type code), so they are demangled here (c++filt, with DF16_ spelled
as the older half-float code Dh) and shortened the way the other rows are."""
import functools
import re
import subprocess


@functools.lru_cache(maxsize=None)
def demangle(name: str) -> str:
    if not name.startswith("_Z"):
        return name
    try:
        out = subprocess.run(["c++filt", name.replace("DF16_", "Dh")], capture_output=True, text=True, timeout=10).stdout.strip()
        return out.replace("__fp16", "f16").replace("_Float16", "f16") if out and not out.startswith("_Z") else name
    except Exception:
        return name


def short(name: str, limit: int = 0) -> str:
    name = demangle(name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("unsigned short", "bf16")
    return name if (name.startswith("zett::") or not limit) else name[:limit]
