"""Builds libhhrl_hip.so (gfx950) in-tree with hipcc.

    python hhrl-vae_amd/build.py [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting shared library travels to the GPU box with the source tree.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libhhrl_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "hhrl_hip.h"))
    return sources() + hdrs


def up_to_date() -> bool:
    if not os.path.exists(LIB_PATH):
        return False
    t = os.path.getmtime(LIB_PATH)
    return all(os.path.getmtime(p) <= t for p in _deps())


def build_lib(force: bool = False, verbose: bool = True) -> str:
    """Compile every .hip source into one shared library.  Objects are built
    in parallel (one hipcc per file), then linked."""
    if not force and up_to_date():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    flags = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and all(os.path.getmtime(p) <= os.path.getmtime(obj)
                                                     for p in [src] + [d for d in _deps() if d.endswith(".h")]):
            continue
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print("FAILED:", src, file=sys.stderr)
            if not verbose:
                print(out, file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
