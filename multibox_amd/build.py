"""Build libmbx.so (HIP, gfx950) in-tree with hipcc.  Used by __graft_entry__.build().

One object per source so that each file gets its own flags; objects are rebuilt only when
the source (or a header) is newer.  The .so is git-ignored but travels to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmbx.so")
ARCH = "gfx950"

# source -> extra flags.  -ffp-contract=off where float rounding sequence is part of parity.
SOURCES = {
    "priors.cpp": ["-ffp-contract=off"],
    "postproc.hip": ["-ffp-contract=off"],
    "conv.hip": ["-munsafe-fp-atomics"],
    # (the atomic optimizer turns the tile-fetch atomicAdd of conv_igemm5_kernel into scan + v_readfirstlane of the result,
    # i.e. waits for it at the issue; without it the wait sits where the value is published)
    "conv5.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"],
    "conv7.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"],        # (as conv5.hip: the tile-fetch atomics)
    "convd.hip": [],
    "convr.hip": [],
    "nnops.hip": ["-munsafe-fp-atomics"],
    "augment.hip": ["-ffp-contract=off"],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]
COMMON += os.environ.get("MBX_BUILD_DEFS", "").split()      # debug builds, e.g. -DMBX_I5_STAMPS (tools/i5_stamps.py); rebuild with force


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libmbx needs ROCm's hipcc to build")


def _newer(a, bs):
    if not os.path.exists(a):
        return True
    ta = os.path.getmtime(a)
    return any(os.path.getmtime(b) > ta for b in bs)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "mbx.h"))
    headers.append(os.path.abspath(__file__))
    objs = []
    procs = []
    for src, flags in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            cmd = [hipcc] + COMMON + flags + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", s, "-o", o]
            if verbose:
                print("[mbx build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append((src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join("== %s ==\n%s" % f for f in failed))
    if force or _newer(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        if verbose:
            print("[mbx build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
