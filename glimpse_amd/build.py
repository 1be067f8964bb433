"""Build libglimpse_hip.so for gfx950 with hipcc (no GPU needed: hipcc cross-compiles).

    python -m glimpse_amd.build            # or glimpse_amd.build.build()

The shared library is written in-tree (glimpse_amd/lib/) so that it travels to the GPU
box with the repository snapshot.  -ffp-contract=off keeps float64 expressions
bit-identical to NumPy's (no implicit FMA); the SSD kernel asks for FMA explicitly.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "glimpse_hip.hip")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libglimpse_hip.so")
DEPS = [
    SRC,
    os.path.join(HERE, "csrc", "glh_kernels.h"),
    os.path.join(HERE, "csrc", "glh_point.h"),
    os.path.join(HERE, "csrc", "glh_math.h"),
    os.path.join(HERE, "csrc", "glh_median.h"),
    os.path.join(HERE, "csrc", "glh_host.h"),
    os.path.join(HERE, "csrc", "glh_comm.h"),
    os.path.join(os.path.dirname(HERE), "include", "glimpse_hip.h"),
]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm)")
    return exe


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build(force=False, verbose=True, extra=()):
    if not force and up_to_date():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc(), *FLAGS, *extra, "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
