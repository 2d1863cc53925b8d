"""Builds liblemon_hip.so (the C-ABI HIP library) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.  Each
translation unit is compiled to an object under lemon_amd/csrc/_obj/ (only when it or a header
changed, several at a time) and the objects are linked into the shared library.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
SO = os.path.join(HERE, "liblemon_hip.so")
SOURCES = ["api.hip", "rowwise.hip", "knn_f32.hip", "knn_bf16.hip", "attention.hip", "linear.hip", "preprocess.hip",
           "gridf1.hip", "dedup.hip", "encoder.hip", "gemm_f16x3.hip"]
HEADERS = ["common.hpp", "knn_common.hpp", "split3.hpp", "scan_plan.hpp", os.path.join("..", "..", "include", "lemon_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: liblemon_hip.so cannot be built")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


def needs_build():
    deps = [os.path.join(CSRC, s) for s in _sources() + HEADERS]
    return _stale(SO, deps)


def build_hip(force=False, verbose=False, jobs=None):
    if not force and not needs_build():
        return SO
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    todo, objs = [], []
    for s in _sources():
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            todo.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(4, len(todo))) as pool:
            list(pool.map(run, todo))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-lhipblaslt"])
    return SO


if __name__ == "__main__":
    import sys
    print(build_hip(force="--force" in sys.argv, verbose=True))
