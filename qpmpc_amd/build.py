"""Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU).

Every translation unit is compiled to its own object (in parallel, only when its
source or a header is newer) and the objects are linked into
``qpmpc_amd/lib/libmpcqp_hip.so``. Objects and the library are git-ignored; the
library travels to the GPU box with the tree.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_UNITS = ("mpcqp_lds.hip", "mpcqp_w64.hip", "mpcqp_pair.hip", "mpcqp_quad.hip", "mpcqp_quadw.hip", "mpcqp_quad4.hip", "mpcqp_quad4w.hip", "mpcqp_big.hip", "mpcqp_bigsolve.hip",
          "mpcqp_model.hip", "mpcqp_stage.hip", "mpcqp_stagew.hip", "mpcqp_stageg.hip", "mpcqp_capi.hip")
SOURCES = [os.path.join(_PKG, "csrc", f) for f in _UNITS]
# units that include another unit's source (mpcqp_quadw.hip compiles the wide instantiations of mpcqp_quad.hip): rebuilt with it
_INCLUDES = {"mpcqp_quadw.hip": ("mpcqp_quad.hip",), "mpcqp_quad4w.hip": ("mpcqp_quad4.hip",)}
# every header a unit may include: the public one and everything under csrc/ (mpcqp_internal.h, mpcqp_plant.h, ...)
HEADERS = [os.path.join(_ROOT, "include", "mpcqp.h")] + sorted(glob.glob(os.path.join(_PKG, "csrc", "*.h")))
LIB_PATH = os.path.join(_PKG, "lib", "libmpcqp_hip.so")
OBJ_DIR = os.path.join(_PKG, "lib", "obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise FileNotFoundError("hipcc not found (set HIPCC=...)")


def _sources():
    return [s for s in SOURCES if os.path.exists(s)]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in _sources() + HEADERS)


def _obj_for(src: str, extra=()) -> str:
    # the flags are part of the object's name: an object built with other flags is never reused
    tag = hashlib.sha1(" ".join([*FLAGS, *extra]).encode()).hexdigest()[:8]
    return os.path.join(OBJ_DIR, f"{os.path.basename(src)}.{tag}.o")


def _compile(src: str, force: bool, verbose: bool, extra) -> str:
    obj = _obj_for(src, extra)
    deps = [src] + HEADERS + [os.path.join(_PKG, "csrc", inc) for inc in _INCLUDES.get(os.path.basename(src), ())]
    if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj
    cmd = [_hipcc(), *FLAGS, *extra, "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_PKG, "csrc"),
           "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    """hipcc --offload-arch=gfx950 -> qpmpc_amd/lib/libmpcqp_hip.so"""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda s: _compile(s, force, verbose, list(extra_flags)), srcs))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


def one_line_command() -> str:
    """The single hipcc invocation that builds the library from its sources (INTEGRATION.md section 1 shows exactly this text;
    tests/test_host_api.py compares the two, so a translation unit added here cannot be forgotten there)."""
    srcs = [f"qpmpc_amd/csrc/{u}" for u in _UNITS]
    lines, cur = [], "     "
    for src in srcs:
        if len(cur) + len(src) + 1 > 104:
            lines.append(cur.rstrip() + " \\")
            cur = "     "
        cur += " " + src
    lines.append(cur + " -o libmpcqp_hip.so")
    return "hipcc " + " ".join(FLAGS) + " -shared -Iinclude -Iqpmpc_amd/csrc \\\n" + "\n".join(lines)


if __name__ == "__main__":
    import sys

    if "--print-command" in sys.argv:
        print(one_line_command())
    else:
        print(build_library(force="--force" in sys.argv, verbose=True))
