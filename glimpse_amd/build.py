"""Build libglimpse_hip.so for gfx950 with hipcc (no GPU needed: hipcc cross-compiles).

    python -m glimpse_amd.build            # or glimpse_amd.build.build()

The shared library is written in-tree (glimpse_amd/lib/) so that it travels to the GPU
box with the repository snapshot.  -ffp-contract=off keeps float64 expressions
bit-identical to NumPy's (no implicit FMA); the SSD kernel asks for FMA explicitly.

The library is many translation units compiled in parallel: glimpse_hip.hip (the C ABI and the
staged kernels) and one object per instantiation of the fused kernel (glh_point_inst.hip with
-DPT_*; the list is csrc/glh_point_variants.h).  Objects are cached in glimpse_amd/lib/obj/ and
rebuilt when a source they include is newer.
"""
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "glimpse_hip.hip")
INST = os.path.join(CSRC, "glh_point_inst.hip")
VARIANTS = os.path.join(CSRC, "glh_point_variants.h")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libglimpse_hip.so")
HEADERS = [
    os.path.join(CSRC, "glh_kernels.h"),
    os.path.join(CSRC, "glh_point.h"),
    os.path.join(CSRC, "glh_math.h"),
    os.path.join(CSRC, "glh_median.h"),
    VARIANTS,
    os.path.join(os.path.dirname(HERE), "include", "glimpse_hip.h"),
]
HOST_HEADERS = [os.path.join(CSRC, "glh_host.h"), os.path.join(CSRC, "glh_comm.h")]
DEPS = [SRC, INST, *HEADERS, *HOST_HEADERS]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
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


def variants():
    """[(TB, PPT, NOBS, SURF, FAST, CON)] from glh_point_variants.h."""
    text = open(VARIANTS).read()
    shapes_line = re.search(r"#define GLH_PT_SHAPES\(X\)(.*)", text).group(1)
    shapes = [tuple(int(v) for v in m) for m in re.findall(r"X\((\d+),\s*(\d+),\s*(\d+)\)", shapes_line)]
    codes_text = text[text.index("#define GLH_PT_CODES"):]
    codes_text = codes_text[: codes_text.index("\n\n")]
    codes = [tuple(int(v) for v in m) for m in re.findall(r"X\(TB, PPT, NOBS,\s*(\d),\s*(\d),\s*(\d)\)", codes_text)]
    if not shapes or not codes:
        raise RuntimeError("could not read the instantiation lists of glh_point_variants.h")
    return [s + c for s in shapes for c in codes]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def up_to_date():
    return not _newer(LIB, DEPS)


def _jobs(extra, objdir=OBJDIR):
    """[(object, command, dependencies)]"""
    cc = hipcc()
    jobs = [(os.path.join(objdir, "glimpse_hip.o"), [cc, *FLAGS, *extra, "-c", SRC], [SRC, *HEADERS, *HOST_HEADERS])]
    for tb, ppt, nobs, s, f, c in variants():
        obj = os.path.join(objdir, f"pt_{tb}_{ppt}_{nobs}_{s}{f}{c}.o")
        defs = [f"-DPT_TB={tb}", f"-DPT_PPT={ppt}", f"-DPT_NOBS={nobs}", f"-DPT_SURF={s}", f"-DPT_FAST={f}",
                f"-DPT_CON={c}"]
        jobs.append((obj, [cc, *FLAGS, *extra, *defs, "-c", INST], [INST, *HEADERS]))
    return jobs


def build(force=False, verbose=True, extra=(), workers=None, out=None):
    """`out`: build an experimental variant (extra compiler flags) under another name, glimpse_amd/lib/<out>.so, with its
    own object directory -- tools/ab.sh runs such libraries side by side (GLH_LIB)."""
    lib, objdir = LIB, OBJDIR
    if out:
        lib, objdir, force = os.path.join(LIBDIR, out + ".so"), os.path.join(LIBDIR, "obj_" + out), True
    if not force and up_to_date():
        return LIB
    os.makedirs(objdir, exist_ok=True)
    jobs = _jobs(list(extra), objdir)
    todo = [(obj, cmd) for obj, cmd, deps in jobs if force or _newer(obj, deps)]
    if workers is None:
        workers = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("GLH_BUILD_JOBS", "8"))))
    if verbose:
        print(f"hipcc: {len(todo)} of {len(jobs)} objects to compile, {workers} at a time", flush=True)

    def run(job):
        obj, cmd = job
        r = subprocess.run([*cmd, "-o", obj], capture_output=True, text=True)
        return obj, r

    with ThreadPoolExecutor(workers) as pool:
        for obj, r in pool.map(run, todo):
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"hipcc failed for {os.path.basename(obj)}")
            if verbose and r.stderr.strip():
                sys.stderr.write(r.stderr)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *[obj for obj, _, _ in jobs]]
    if verbose:
        print(" ".join(cmd[:6]) + f" ... ({len(jobs)} objects)", flush=True)
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
