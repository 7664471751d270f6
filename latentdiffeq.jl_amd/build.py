"""Compile liblde.so (HIP kernels + C ABI) for gfx950, in-tree.

    python latentdiffeq.jl_amd/build.py [--force] [-v]     # or: import latentdiffeq_amd; latentdiffeq_amd.build_lib()

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so that it travels with
the source tree (it is git-ignored, not gpurun-ignored). Every source is compiled to its own object
(in parallel, only when it or a header changed), then linked: touching one kernel file rebuilds in seconds.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "liblde.so")
SOURCES = ["lde_api.hip", "lde_pendulum.hip", "lde_mlp.hip", "lde_chain.hip", "lde_rnn.hip", "lde_loss.hip", "lde_optim.hip", "lde_comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "lde.h")]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False, extra_flags=(), out: str = LIB) -> str:
    """Build (if stale) and return the library path. `extra_flags` (e.g. ["-DLDE_PROF=1"]) with another `out` makes a
    diagnostic build next to the product one (objects are then kept in a directory of their own)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = OBJ if out == LIB else out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _newer(obj, [src, __file__] + hdrs):
            cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        return cmd, subprocess.run(cmd, capture_output=True, text=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            for cmd, r in ex.map(run, jobs):
                if r.returncode != 0:
                    sys.stderr.write(r.stdout + r.stderr)
                    raise RuntimeError("hipcc failed: " + " ".join(cmd))
                if verbose:
                    sys.stderr.write(r.stderr)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    # k_mlpb / k_mlpc keep their weight-gradient tiles in AGPRs the compiler is not told about (csrc/lde_mlpb.h): the object is checked —
    # the compiler's own code must not touch that range — on EVERY build_lib() call whose object has no stamp of a passed check (the stamp
    # names the hipcc that compiled it). A violation, or a check that cannot run, removes the object: never a linked, unverified library.
    mlp_obj = os.path.join(objdir, "lde_mlp.o")
    stamp = mlp_obj + ".checked"
    if _newer(stamp, [mlp_obj]):
        try:
            try:
                from .check_agprs import check_object
            except ImportError:      # run as a script: no parent package
                sys.path.insert(0, HERE)
                from check_agprs import check_object
            bad = check_object(mlp_obj)
        except Exception as e:       # llvm-objdump missing, unreadable object, …: the check did not run
            bad = [f"the check could not run: {e!r}"]
        if bad:
            for f in (mlp_obj, stamp, out):
                if os.path.exists(f):
                    os.remove(f)
            raise RuntimeError("lde_mlp.o: the compiler uses hidden accumulator registers of k_mlpb / k_mlpc:\n  " + "\n  ".join(bad[:12]))
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().splitlines()
        with open(stamp, "w") as f:
            f.write("hidden AGPR ranges respected\n" + "\n".join(ver[:2]) + "\n")
    # what was validated travels in the library itself: lde_build_info() returns the stamp's text (compiler version + the check's verdict)
    info_src, info_obj = os.path.join(objdir, "lde_buildinfo.cpp"), os.path.join(objdir, "lde_buildinfo.o")
    text = open(stamp).read().strip().replace("\\", "/").replace('"', "'").replace("\n", "; ")
    body = f'extern "C" __attribute__((visibility("default"))) const char* lde_build_info(void) {{ return "{text}"; }}\n'
    if not os.path.exists(info_src) or open(info_src).read() != body or not os.path.exists(info_obj):
        with open(info_src, "w") as f:
            f.write(body)
        r = subprocess.run([shutil.which("g++") or hipcc, "-O1", "-fPIC", "-c", info_src, "-o", info_obj], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("compiling lde_buildinfo.cpp failed")
    objs.append(info_obj)
    if jobs or _newer(out, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed linking liblde.so")
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose="-v" in sys.argv))
