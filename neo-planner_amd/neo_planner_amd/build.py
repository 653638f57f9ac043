"""Builds libneo_planner_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(PKG), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(PKG)), "include")
LIB = os.path.join(PKG, "libneo_planner_hip.so")
SOURCES = ["neo_kernels.hip"]
DEPS = ["neo_kernels.hip", "neo_device.hpp", "neo_lbfgs.hpp", "neo_linesearch.hpp", "neo_lbfgs_sm.hpp", "neo_group.hpp",
        "neo_group_kernel.hpp"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, d) for d in DEPS] + [os.path.join(INCLUDE, "neo_planner.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """compile the HIP library if it is missing or older than its sources; returns its path"""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libneo_planner_hip.so")
    # NEO_BUILD_DEFS="-DNEO_STAMPS ..." : experiment builds only (tools/); the product is built without
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           "-I", INCLUDE] + os.environ.get("NEO_BUILD_DEFS", "").split() + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
