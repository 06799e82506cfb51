"""Builds the HIP shared library in-tree (x264_amd/libx264hip.so) for gfx950.  hipcc cross-compiles
without a GPU, so this runs on the build host; the .so travels to the GPU box with the snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "x264hip.hip"), os.path.join(HERE, "csrc", "lookahead_host.cpp"), os.path.join(HERE, "csrc", "shard_host.cpp")]
import glob
HDR = sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
OUT = os.path.join(HERE, "libx264hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in SRC + HDR)


def build_profile(verbose=False):
    """x264_amd/libx264hip_prof.so: the same library with cycle counters compiled into the search kernel (-DME_PROFILE);
    selected at run time with X264HIP_LIB=<path> (measurement aid, never the default)."""
    out = os.path.join(HERE, "libx264hip_prof.so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DME_PROFILE", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(HERE, "csrc"), "-o", out] + SRC + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_variant(tag, defines, verbose=False):
    """x264_amd/libx264hip_<tag>.so built with extra -D flags: A/B experiments on the GPU box, selected with X264HIP_LIB=<path>."""
    out = os.path.join(HERE, "libx264hip_%s.so" % tag)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"] + ["-D" + d for d in defines] + \
          ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"), "-o", out] + SRC + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


OBJDIR = os.path.join(HERE, "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def _deps(src):
    """what an object is rebuilt for: its source, the public header, and -- for the device translation unit -- every kernel header"""
    pub = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    return [src] + pub + (sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) if src.endswith(".hip") else [])


def build(force=False, verbose=False):
    """One object per source under x264_amd/_obj (the device translation unit takes minutes, the host ones seconds: only what changed is
    compiled again), linked into x264_amd/libx264hip.so."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJDIR, exist_ok=True)
    # -ffp-contract=off: the few FP32 expressions on this path (AQ, MB-tree) must round like the reference's
    # separate multiply/add instructions; hipcc's default would fuse them into FMAs (1-ulp differences)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]
    objs, jobs = [], []
    for src in SRC:
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src)):
            cmd = [hipcc] + FLAGS + inc + ["-x", "hip", "-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait():
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if jobs or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", OUT] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m x264_amd.build --variant <tag> DEF1 DEF2=3 ...
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True)
    elif "--prof" in sys.argv:
        build_profile(verbose=True)
    else:
        build(force="--force" in sys.argv, verbose=True)
