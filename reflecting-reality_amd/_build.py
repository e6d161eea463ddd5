"""Builds libmfhip.so (the gfx950 kernel library) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container; the
resulting ``reflecting-reality_amd/lib/libmfhip.so`` travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG_DIR, "csrc")
LIB_DIR = os.path.join(_PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmfhip.so")
INCLUDE = os.path.join(os.path.dirname(_PKG_DIR), "include")

SOURCES = ["gemm_conv.hip", "norm.hip", "attention.hip", "elementwise.hip", "train.hip", "frontend.hip", "fp8.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
               f"-I{INCLUDE}"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmfhip.so cannot be built on this machine")


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    with open(os.path.join(INCLUDE, "mfhip.h"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def is_fresh() -> bool:
    stamp = LIB_PATH + ".sha256"
    if not (os.path.exists(LIB_PATH) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == _digest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip source for gfx950 and link libmfhip.so. Returns the library path."""
    if not force and is_fresh():
        return LIB_PATH
    hipcc = _hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [hipcc, *HIPCC_FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[mfhip build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    if verbose:
        print("[mfhip build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    with open(LIB_PATH + ".sha256", "w") as f:
        f.write(_digest())
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
