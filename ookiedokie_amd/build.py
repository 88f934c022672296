"""Builds libookiedokie_amd.so (hand-written gfx950 kernels + C-ABI) in-tree.

    python -m ookiedokie_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off is part
of the contract: the kernels spell out every fused multiply-add themselves
(see csrc/kernels.hip).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libookiedokie_amd.so")
SOURCES = ["loaders.cpp", "rx.cpp", "synth.cpp", "stream_fir.cpp", "backend.cpp", "edges_fsm.hip", "fsm_scan.hip",
           "formatter.cpp", "kernels.hip", "fir_mfma.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(HERE, "..", "include", "ookiedokie_amd.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    for src in sources():
        obj = os.path.join(HERE, "lib", os.path.basename(src) + ".o")
        cmd = [_hipcc(), "-x", "hip", "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
               "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-result", "-Wno-uninitialized",
               "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
               "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr:
            print(r.stderr)
        objs.append(obj)
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
