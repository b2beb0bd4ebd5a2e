"""Build the in-tree native libraries (hipcc for gfx950, gcc for the C host side).

    python -m hercules_amd.build [--force]

libhq_solver.so : HIP kernels + C-ABI (include/hq_solver.h)
libhq_solver_f32.so : the same sources with -DHQ_SINGLE_PRECISION_SOLVER (hq_real = float; never the default)
libhq_host.so   : C host side mirroring the reference's solver_init/solver_run
                  for uniform boxes (include/hq_host.h), links libhq_solver.so
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOLVER_LIB = os.path.join(CSRC, "libhq_solver.so")
HOST_LIB = os.path.join(CSRC, "libhq_host.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_solver(force=False):
    srcs = [os.path.join(CSRC, f) for f in ("hq_engine.hip", "hq_kernels.h", "hq_opts.h", "hq_patch.h", "hq_brick.h")]
    srcs.append(os.path.join(ROOT, "include", "hq_solver.h"))
    if force or _newer(SOLVER_LIB, srcs):
        cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
               "-fvisibility=hidden", "-fopenmp", "-Wno-unused-value"] + \
              os.environ.get("HQ_EXTRA_FLAGS", "").split() + \
              ["-o", SOLVER_LIB, srcs[0], "-Wl,-rpath,/opt/rocm/lib", "-Wl,-Bsymbolic", "-ldl"]
        subprocess.check_call(cmd, cwd=CSRC)
    return SOLVER_LIB


SOLVER_LIB_F32 = os.path.join(CSRC, "libhq_solver_f32.so")


def build_solver_f32(force=False):
    """The same sources with -DHQ_SINGLE_PRECISION_SOLVER: hq_real = float (the reference's -DSINGLE_PRECISION_SOLVER,
    psolve.h:60-64) -- a separately named library, never the default."""
    srcs = [os.path.join(CSRC, f) for f in ("hq_engine.hip", "hq_kernels.h", "hq_opts.h", "hq_patch.h", "hq_brick.h")]
    srcs.append(os.path.join(ROOT, "include", "hq_solver.h"))
    if force or _newer(SOLVER_LIB_F32, srcs):
        cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-DHQ_SINGLE_PRECISION_SOLVER",
               "-fvisibility=hidden", "-fopenmp", "-Wno-unused-value"] + \
              os.environ.get("HQ_EXTRA_FLAGS", "").split() + \
              ["-o", SOLVER_LIB_F32, srcs[0], "-Wl,-rpath,/opt/rocm/lib", "-Wl,-Bsymbolic", "-ldl"]
        subprocess.check_call(cmd, cwd=CSRC)
    return SOLVER_LIB_F32


def build_host(force=False):
    src = os.path.join(CSRC, "hq_host.c")
    if not os.path.exists(src):
        return None
    deps = [src, os.path.join(CSRC, "hq_mesher.h"), os.path.join(ROOT, "include", "hq_host.h"),
            os.path.join(ROOT, "include", "hq_solver.h")]
    if force or _newer(HOST_LIB, deps):
        cmd = ["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-fvisibility=hidden", "-fopenmp",
               "-I", os.path.join(ROOT, "include"), "-o", HOST_LIB, src,
               "-L", CSRC, "-lhq_solver", "-Wl,-rpath,$ORIGIN", "-lm"]
        subprocess.check_call(cmd, cwd=CSRC)
    return HOST_LIB


EXAMPLE_BIN = os.path.join(ROOT, "examples", "hq_psolve_mini")


def build_example(force=False):
    """The all-C host program (examples/hq_psolve_mini.c) on the two libraries."""
    src = os.path.join(ROOT, "examples", "hq_psolve_mini.c")
    if not os.path.exists(src):
        return None
    if force or _newer(EXAMPLE_BIN, [src, SOLVER_LIB, HOST_LIB]):
        cmd = ["gcc", "-O2", "-std=gnu99", "-I", os.path.join(ROOT, "include"), "-o", EXAMPLE_BIN, src,
               "-L", CSRC, "-lhq_host", "-lhq_solver", "-Wl,-rpath,$ORIGIN/../hercules_amd/csrc",
               "-Wl,-rpath,/opt/rocm/lib", "-lm"]
        subprocess.check_call(cmd)
    return EXAMPLE_BIN


def build(force=False):
    out = build_solver(force), build_host(force)
    build_solver_f32(force)
    build_example(force)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
