"""Build libkmu.so (HIP, gfx950) in-tree: kmerutils_amd/libkmu.so.  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libkmu.so")
SOURCES = ["kmu_api.hip", "kmu_sketch.hip", "kmu_sketch_kernels.hip", "kmu_sketch_super.hip", "kmu_sketch_dens.hip", "kmu_count.hip", "kmu_count_part.hip", "kmu_count_dist.hip", "kmu_smer.hip", "kmu_hostpack.hip", "kmu_compare.hip",
           "kmu_ingest.hip", "kmu_kmergen.hip", "kmu_comm.hip"]
HEADERS = ["kmu_device.h", "kmu_stream.h", "kmu_ctx.hpp", "kmu_comm.hpp", "kmu_flat.h", "kmu_count_table.h", "kmu_sketch_kernels.h", "kmu_smer.h", "kmu_smer.hpp", "kmu_hostpack.hpp", os.path.join("..", "..", "include", "kmu.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [_hipcc()] + FLAGS + os.environ.get("KMU_BUILD_DEFS", "").split() + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed on %s" % s)
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl", "-lpthread"]
    subprocess.check_call(cmd)
    return OUT


# ---- host side above the C-ABI (C++, g++): the two tools ----------------------------------------------------------------
ROOT = os.path.dirname(HERE)
BIN = os.path.join(HERE, "bin")
HOST_PROGRAMS = {
    os.path.join(BIN, "datasketcher"): os.path.join(HERE, "tools", "datasketcher.cpp"),
    os.path.join(BIN, "parsefastq"): os.path.join(HERE, "tools", "parsefastq.cpp"),
}


def compile_host(src, out, extra=(), force=False, verbose=False, deps=()):
    """g++ one host program against libkmu.so (rpath relative to the program, so the tree can move)"""
    hdr = os.path.join(ROOT, "include", "kmerutils.hpp")
    deps = [src, hdr, OUT] + list(deps)
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-pthread", src, "-o", out, "-L" + HERE, "-lkmu",
           "-Wl,-rpath,$ORIGIN/" + os.path.relpath(HERE, os.path.dirname(out)), "-Wl,-rpath-link,/opt/rocm/lib"] + list(extra)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_host(force=False, verbose=False):
    build()
    return [compile_host(src, out, force=force, verbose=verbose) for out, src in HOST_PROGRAMS.items()]


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
    for o in build_host(force="--force" in sys.argv, verbose=True):
        print(o)
