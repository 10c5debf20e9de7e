"""Build libppbo_hip.so (gfx950) in-tree with hipcc.

    python -m ppbo_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the gpurun snapshot; nothing is JIT-compiled at import time.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libppbo_hip.so")
STAMP = os.path.join(HERE, ".libppbo_hip.stamp")
SOURCES = ["capi.hip", "gram.hip", "gemm.hip", "linalg.hip", "fit.hip", "predict.hip", "fused.hip", "meangrad.hip", "rff.hip", "lu.hip", "dist.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fvisibility=hidden"]   # the dynamic symbol table is include/ppbo_hip.h (PPBO_API), nothing else


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; tried $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _digest():
    h = hashlib.sha256()
    names = sorted(os.listdir(CSRC)) + ["../../include/ppbo_hip.h"]
    for n in names:
        pth = os.path.join(CSRC, n)
        if os.path.isfile(pth):
            h.update(n.encode())
            with open(pth, "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        procs.append((s, subprocess.Popen([hipcc, *FLAGS, "-c", s, "-o", o], stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True)))
    for s, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC",
           "-Wl,--version-script=" + os.path.join(CSRC, "libppbo_hip.map"), "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    with open(STAMP, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
