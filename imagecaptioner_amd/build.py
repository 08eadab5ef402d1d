"""In-tree build of libtagpipe_hip.so (HIP kernels + C ABI) for gfx950.

hipcc cross-compiles without a GPU.  The library is written next to the
sources so that it travels with the tree; it is never installed elsewhere.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtagpipe_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))

SOURCES = ["tp_kernels.hip", "tp_api.cpp", "tp_tables.cpp"]
HEADERS = ["tp_common.h", "tp_pipeline.h", "tp_fast.h", "tp_lean.h", "tp_launch.h", "tp_tables.hpp", "tp_json.hpp", "tp_case_tables.inc",
           os.path.join("..", "..", "include", "tagpipe.h")]
# -ffp-contract=off: the score arithmetic (1.0 - i * 0.01) must round like the reference's, with no fused multiply-add.
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, prof=False):
    """prof=True builds the diagnostic variant libtagpipe_hip_prof.so (per-section cycle counters inside k_fast)."""
    if prof:
        return _build(LIB.replace(".so", "_prof.so"), "_prof.o", ["-DTP_PROF"], verbose)
    variant = os.environ.get("TAGPIPE_BUILD_VARIANT")          # experiments: "name:-Dflag,-Dflag" -> libtagpipe_hip_name.so
    if variant:
        name, flags = variant.split(":", 1)
        return _build(LIB.replace(".so", "_%s.so" % name), "_%s.o" % name, [f for f in flags.split(",") if f], verbose, kernels_only=True)
    if not force and not _stale():
        return LIB
    return _build(LIB, ".o", [], verbose)


def _build(lib, obj_suffix, extra, verbose, kernels_only=False):
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + obj_suffix)
        if kernels_only and not src.endswith(".hip"):       # experiments only touch the kernels: reuse the main build's host objects
            objs.append(os.path.join(CSRC, os.path.splitext(src)[0] + ".o"))
            continue
        cmd = [HIPCC, "--offload-arch=gfx950"] + COMMON + extra
        if src.endswith(".hip"):
            cmd += ["-x", "hip"]
        else:
            cmd += ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include")]
        cmd += ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, prof="--prof" in sys.argv)
