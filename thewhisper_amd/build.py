"""Builds thewhisper_amd/lib/libthewhisper_gfx950.so from csrc/*.hip with hipcc for gfx950 (MI355X only).

The library is plain HIP (no torch headers): `hipcc --offload-arch=gfx950 -O3 -shared -fPIC`.
hipcc cross-compiles without a GPU, so this also is the CPU-side "does it build" check.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libthewhisper_gfx950.so"
SOURCES = ["api.hip", "k_gemm.hip", "k_misc.hip", "k_logmel.hip", "k_attn.hip", "k_decode.hip", "k_dtw.hip", "k_vad.hip"]
# -amdgpu-kernarg-preload-count: gfx950 hands the leading kernel-argument dwords to every wave in SGPRs (as many as the free user
# SGPRs allow); the decode kernels order their arguments so that the request addresses need nothing else (k_decode.hip)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built (there is no CPU fallback)")


def _digest(paths: List[str]) -> str:
    h = hashlib.sha256()
    for p in sorted(paths, key=os.path.basename):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())   # names, not paths: the digest must be the same wherever the tree is checked out
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def source_digest() -> str:
    """sha256 over every kernel source + header + the flags: identifies WHICH kernels a measurement belongs to (bench.py
    refuses a PMC traffic summary taken with other kernels)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    return _digest(srcs + [os.path.join(CSRC, "tw_common.h"), os.path.join(HERE, "..", "include", "thewhisper.h")])


def library_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "tw_common.h"), os.path.join(HERE, "..", "include", "thewhisper.h")]
    stamp = os.path.join(LIBDIR, "build.sha256")
    want = _digest(deps)
    out = library_path()
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return out
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(want)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
