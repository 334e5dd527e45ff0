"""Build libcalhip.so (gfx950 only) in-tree with hipcc.

``python -m cal_amd.build`` or ``cal_amd.build.build()``.  Objects and the
shared library land in ``cal_amd/lib/`` (git-ignored, but they travel to the
GPU box with the gpurun snapshot).  Sources are recompiled only when newer than
their object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcalhip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-ffp-contract=on", "-munsafe-fp-atomics"]
FLAGS += os.environ.get("CAL_HIPCC_EXTRA", "").split()      # e.g. -DCAL_RO_CLOCKS (profiling builds)


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libcalhip.so cannot be built")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def kernel_source_sha() -> str:
    """sha1 (16 hex digits) over every HIP source and header of libcalhip.so: identifies the kernel table a measurement
    (a PMC summary under profiles/, a bench line) belongs to."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _needs(src: str, obj: str, deps) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


HOST_SRC = os.path.join(HERE, "csrc_host", "calhost.cpp")
HOST_LIB = os.path.join(LIBDIR, "libcalhost.so")


def build_host(force: bool = False, verbose: bool = True) -> str:
    """libcalhost.so: the plain-C++ HOST implementation of the operator-level entry points (same symbols; g++, no HIP)."""
    os.makedirs(LIBDIR, exist_ok=True)
    hdr = os.path.join(os.path.dirname(HERE), "include", "cal_hip.h")
    if force or _needs(HOST_SRC, HOST_LIB, [hdr]):
        cxx = shutil.which("g++") or shutil.which("c++")
        if not cxx:
            raise RuntimeError("g++ not found: libcalhost.so cannot be built")
        cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-fopenmp", "-Wall",
               "-I", os.path.join(os.path.dirname(HERE), "include"), HOST_SRC, "-o", HOST_LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host build failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[cal_amd.build] built", HOST_LIB, flush=True)
    return HOST_LIB


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    build_host(force, verbose)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "cal_hip.h"))
    headers = [h for h in headers if os.path.exists(h)]
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _needs(src, obj, headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-I", os.path.join(os.path.dirname(HERE), "include"), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose:
            print("[cal_amd.build] compiled", os.path.basename(src), flush=True)
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        warns = list(ex.map(cc, jobs))
    for w in warns:
        if w and verbose:
            sys.stderr.write(w)
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[cal_amd.build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
