"""Build libontovae_hip.so in-tree with hipcc for gfx950 (no JIT, no torch extension machinery)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libontovae_hip.so")
SOURCES = ["ontovae_plan.cpp", "ontovae_kernels.hip", "ontovae_stats.hip"]
HEADERS = ["ontovae_internal.h", "ontovae_device.h", "ontovae_gemm.h", "ontovae_bf16.h", "ontovae_dense.h", os.path.join("..", "..", "include", "ontovae_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    # the .cpp is host-only code; compile it as HIP so one command produces the library
    cmd.insert(1, "-x")
    cmd.insert(2, "hip")
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
