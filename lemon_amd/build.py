"""Builds liblemon_hip.so (the C-ABI HIP library) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "liblemon_hip.so")
SOURCES = ["api.hip", "rowwise.hip", "knn_f32.hip", "knn_bf16.hip", "attention.hip", "linear.hip", "preprocess.hip", "gridf1.hip"]
HEADERS = ["common.hpp", os.path.join("..", "..", "include", "lemon_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: liblemon_hip.so cannot be built")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(p) > t for p in deps)


def build_hip(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-o", SO] + [os.path.join(CSRC, s) for s in SOURCES] + ["-lhipblaslt"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
