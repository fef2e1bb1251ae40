"""Builds libdvbs2hip.so (hand-written HIP for gfx950) in-tree: dvbs2_amd/lib/libdvbs2hip.so.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container
(__graft_entry__.build()).  The built .so is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libdvbs2hip.so")
ARCH = "gfx950"
FLAGS = (os.environ.get("DVBS2HIP_EXTRA_FLAGS", "").split()) + ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libdvbs2hip.so cannot be built (there is no CPU fallback)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(_HERE, "..", "include", "dvbs2hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    for stale in set(glob.glob(os.path.join(LIBDIR, "*.hip.o"))) - {os.path.join(LIBDIR, os.path.basename(src) + ".o") for src in sources()}:
        os.remove(stale)          # the object of a translation unit that no longer exists
    objs = []
    cc = _hipcc()
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [cc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [cc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB + ".tmp"] + objs
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
