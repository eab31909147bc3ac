"""Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SOURCES = [os.path.join(_PKG, "csrc", f) for f in ("mpcqp_lds.hip", "mpcqp_w64.hip", "mpcqp_big.hip", "mpcqp_bigsolve.hip", "mpcqp_model.hip", "mpcqp_capi.hip")]
HEADERS = [os.path.join(_ROOT, "include", "mpcqp.h"), os.path.join(_PKG, "csrc", "mpcqp_internal.h")]
LIB_PATH = os.path.join(_PKG, "lib", "libmpcqp_hip.so")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise FileNotFoundError("hipcc not found (set HIPCC=...)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in SOURCES + HEADERS)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> qpmpc_amd/lib/libmpcqp_hip.so"""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(s)]
    cmd = [
        _hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_PKG, "csrc"),
        *srcs, "-o", LIB_PATH,
    ]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
