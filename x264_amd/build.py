"""Builds the HIP shared library in-tree (x264_amd/libx264hip.so) for gfx950.  hipcc cross-compiles
without a GPU, so this runs on the build host; the .so travels to the GPU box with the snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "x264hip.hip"), os.path.join(HERE, "csrc", "lookahead_host.cpp")]
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
           "-I" + os.path.join(HERE, "csrc"), "-o", out] + SRC
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_variant(tag, defines, verbose=False):
    """x264_amd/libx264hip_<tag>.so built with extra -D flags: A/B experiments on the GPU box, selected with X264HIP_LIB=<path>."""
    out = os.path.join(HERE, "libx264hip_%s.so" % tag)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"] + ["-D" + d for d in defines] + \
          ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"), "-o", out] + SRC
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [s for s in SRC if os.path.exists(s)]
    # -ffp-contract=off: the few FP32 expressions on this path (AQ, MB-tree) must round like the reference's
    # separate multiply/add instructions; hipcc's default would fuse them into FMAs (1-ulp differences)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(HERE, "csrc"), "-o", OUT] + srcs
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
