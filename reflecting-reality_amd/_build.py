"""Builds libmfhip.so (the gfx950 kernel library) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container; the
resulting ``reflecting-reality_amd/lib/libmfhip.so`` travels to the GPU box with the repo snapshot.

The GEMM / conv family is one kernel template (csrc/gemm_conv_kernel.h) instantiated for ~140 (tile, precision)
combinations: each group of tiles is its own translation unit so that the library builds in parallel, and an object is
recompiled only when its source, a header or the flags changed.  Developer variants (``build(variant="stamps",
extra_flags=["-DMF_STAMPS=1"])``) go to ``lib/libmfhip_<variant>.so`` and are selected with MFHIP_LIB; the product
library is always the plain build.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Sequence

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG_DIR, "csrc")
LIB_DIR = os.path.join(_PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmfhip.so")
INCLUDE = os.path.join(os.path.dirname(_PKG_DIR), "include")

# longest compiles first (the pool takes them in this order)
SOURCES = ["train.hip", "gemm_f32_b.hip", "gemm_f32_a.hip", "gemm_f16x3.hip", "gemm_bf16x3.hip", "gemm_bf16_ws_ring.hip", "gemm_f16_ws_ring.hip",
           "gemm_bf16_b.hip", "gemm_f16_b.hip", "gemm_bf16_a.hip", "gemm_bf16_c.hip", "gemm_f16_c.hip", "gemm_bf16_ws_dx.hip", "gemm_f16_ws_dx.hip",
           "gemm_f16_a.hip", "gemm_f16x3_ws.hip", "attention.hip", "gemm_fp8.hip", "conv_halo.hip", "gemm_nloop.hip", "gemm_pers.hip", "gemm_conv.hip", "norm.hip", "elementwise.hip",
           "frontend.hip", "fp8.hip", "program.hip"]
# -pragma-unroll-threshold: the epilogue loops of the GEMM family index their accumulator arrays by the loop counter, so a loop the
# optimizer declines to unroll ("unrolled size is too large", default 16 K instructions) sends 64-160 accumulators per lane to scratch
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
               "-mllvm", "-pragma-unroll-threshold=1000000", f"-I{INCLUDE}"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmfhip.so cannot be built on this machine")


def _headers_digest() -> "hashlib._Hash":
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(".h"):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    with open(os.path.join(INCLUDE, "mfhip.h"), "rb") as f:
        h.update(f.read())
    return h


def _obj_digest(src: str, flags: Sequence[str]) -> str:
    h = _headers_digest()
    with open(os.path.join(CSRC, src), "rb") as f:
        h.update(src.encode())
        h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def _digest(flags: Sequence[str] = HIPCC_FLAGS) -> str:
    h = _headers_digest()
    for name in SOURCES:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def lib_path(variant: Optional[str] = None) -> str:
    return LIB_PATH if not variant else os.path.join(LIB_DIR, f"libmfhip_{variant}.so")


def is_fresh(variant: Optional[str] = None, extra_flags: Sequence[str] = ()) -> bool:
    path = lib_path(variant)
    stamp = path + ".sha256"
    if not (os.path.exists(path) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == _digest([*HIPCC_FLAGS, *extra_flags])


def build(force: bool = False, verbose: bool = True, variant: Optional[str] = None, extra_flags: Sequence[str] = (),
          jobs: Optional[int] = None) -> str:
    """Compile every .hip source for gfx950 and link the library. Returns its path."""
    flags = [*HIPCC_FLAGS, *extra_flags]
    path = lib_path(variant)
    if not force and is_fresh(variant, extra_flags):
        return path
    hipcc = _hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj" if not variant else f"obj_{variant}")
    os.makedirs(obj_dir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        want = _obj_digest(src, flags)
        stamp = obj + ".sha256"
        if not force and os.path.exists(obj) and os.path.exists(stamp):
            with open(stamp) as f:
                if f.read().strip() == want:
                    return obj
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[mfhip build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(want)
        return obj

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", path, *objs]
    if verbose:
        print("[mfhip build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    with open(path + ".sha256", "w") as f:
        f.write(_digest(flags))
    return path


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--variant", default=None, help="developer build: lib/libmfhip_<variant>.so (select it with MFHIP_LIB)")
    ap.add_argument("-D", dest="defs", action="append", default=[], help="extra -D definitions of a developer variant")
    a = ap.parse_args()
    print(build(force=a.force, variant=a.variant, extra_flags=[f"-D{d}" for d in a.defs]))
