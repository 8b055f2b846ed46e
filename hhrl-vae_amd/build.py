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


def build_lib(force: bool = False, verbose: bool = True, defines=(), tag: str = "") -> str:
    """Compile every .hip source into one shared library.  Objects are built
    in parallel (one hipcc per file), then linked.  `defines` + `tag` give a
    diagnostic variant (lib/libhhrl_hip_<tag>.so) of the same sources."""
    if tag:
        return _build(True, verbose, list(defines), tag)
    if not force and up_to_date():
        return LIB_PATH
    return _build(force, verbose, [], "")


def _build(force: bool, verbose: bool, defines, tag: str) -> str:
    lib_path = LIB_PATH if not tag else os.path.join(LIB_DIR, "libhhrl_hip_%s.so" % tag)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(HERE, "build" + ("_" + tag if tag else ""))
    os.makedirs(obj_dir, exist_ok=True)
    flags = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]
    flags += ["-D" + d for d in defines]
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
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib_path


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build_lib(defines=["HHRL_STAMPS"], tag="stamps"))
    else:
        print(build_lib(force="--force" in sys.argv))
