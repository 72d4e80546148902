"""Build libperceptor_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libperceptor_hip.so")
SOURCES = ["igemm.hip", "conv3x3.hip", "conv_wd.hip", "gemm_wd.hip", "norm.hip", "attn.hip", "attn_flash.hip", "elementwise.hip", "clip.hip", "f32gemm.hip", "sampling.hip", "backward.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
# per-file additions.  conv_wd.hip: no SLP vectorisation -- the vectoriser packs the patch staging's f32 arithmetic into v_pk_fma_f32 /
# v_pk_mul_f32, which cost MORE vector-issue time beside MFMAs than the two scalar ops they replace (MI355X_MICROARCH.md, per-instruction
# constants) and need ~800 v_mov to pair registers: 128 -> 128 @512x512 0.715 -> 0.694 ms, c5 bf16 42.39 -> 42.13 ms (same-box A/B)
FILE_FLAGS = {"conv_wd.hip": ["-fno-slp-vectorize"]}


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=()) -> str:
    """variant / extra_flags: diagnostic builds for tools/ (e.g. variant="stamps", extra_flags=["-DPMI_STAMPS"]) written next
    to the product library as libperceptor_hip_<variant>.so; the product build takes neither."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = LIB if not variant else LIB.replace(".so", f"_{variant}.so")
    bdir = "build" if not variant else f"build_{variant}"
    hdrs = [os.path.join(HERE, "common.h"), os.path.join(HERE, "..", "..", "include", "perceptor_hip.h")]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    objs = []

    def compile_one(src):
        obj = os.path.join(HERE, bdir, src.replace(".hip", ".o"))
        os.makedirs(os.path.dirname(obj), exist_ok=True)
        if force or _stale(obj, [os.path.join(HERE, src)] + hdrs):
            cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src, []), *extra_flags, "-c", os.path.join(HERE, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(7, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, variant="stamps", extra_flags=["-DPMI_STAMPS"]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
