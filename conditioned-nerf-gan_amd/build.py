"""Builds libcnerf_hip.so (gfx950) in-tree with hipcc.  Usage: python -m <package>.build  or  build_library()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcnerf_hip.so")
SOURCES = ["cnerf_abi.hip", "field_kernel.hip", "field_h3.hip", "field_pw16.hip", "ray_kernels.hip", "grad_kernels.hip", "bwd16.hip", "chain_pw16.hip", "scatter_patch.hip"]
HEADERS = ["cnerf_dev.hpp", "cnerf_kernels.hpp", "field_common.hpp", "bwd16.hpp", "h3_dev.hpp", os.path.join("..", "..", "include", "cnerf.h")]
# -ffp-contract=off: a*b+c is two roundings unless fmaf() is written (see csrc/cnerf_dev.hpp)
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, extra_flags=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra_flags = tuple(extra_flags) + tuple(os.environ.get("CNERF_EXTRA_FLAGS", "").split())
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    # (source, object, extra defines): field_h3.hip and field_pw16.hip are compiled twice -- fp16x3 split kernels and the single-pass fp16 kernels
    h1_defs = ("-DCNERF_H3_PARTS=1",) + tuple(os.environ.get("CNERF_H1_FLAGS", "").split())      # (experiments: e.g. -DCNERF_H3_OCC=2)
    units = [(src, src.replace(".hip", ".o"), ()) for src in SOURCES] + [("field_h3.hip", "field_h1.o", h1_defs), ("field_pw16.hip", "field_pw1.o", h1_defs)]
    for src, obj, defs in units:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, obj)
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, *FLAGS, *defs, *extra_flags, "-c", s, "-o", o])
    if jobs:      # the translation units are independent: compile them side by side (a handful of processes)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)

        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as pool:
            list(pool.map(run, jobs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
