"""Builds the in-tree native library ``smcp_amd/libsmcp_amd.so`` (HIP kernels + C-ABI) for gfx950.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the
repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsmcp_amd.so")
SOURCES = ["capi.hip", "symbolic.cpp"]
DEPS = ["front_large.hip", "front_mfma.hip", "capi.hip", "kkt.hip", "kkt_qr.hip", "front_generic.hip", "wgblas.hpp", "context.hpp", "symbolic.cpp",
        "symbolic.hpp", "../../include/smcp_amd.h"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for d in os.listdir(CSRC):
        if os.path.getmtime(os.path.join(CSRC, d)) > t:
            return True
    return os.path.getmtime(os.path.join(HERE, "..", "include", "smcp_amd.h")) > t


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-Wno-unused-value", "-pthread", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if os.environ.get("SMCP_STAMPS") == "1":      # diagnostic build with in-kernel cycle stamps
        cmd.insert(1, "-DSMCP_STAMPS")
    for flag in os.environ.get("SMCP_CXXFLAGS", "").split():      # experiment builds (-DSMCP_FAM2_NT ...)
        cmd.insert(1, flag)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
