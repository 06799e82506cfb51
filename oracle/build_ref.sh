#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- builds the *real* reference C path (no asm) into oracle/_ref/.
#
# The reference sources are compiled where they lie under $X264_REF (default /root/reference);
# nothing is copied into this repository and only build outputs are written, all of them under
# oracle/_ref/ (git-ignored, but shipped to the GPU box with the gpurun snapshot).
#
# What runs:
#   1. the reference's own `configure` script, out-of-tree in oracle/_ref/cfg, only to emit the
#      generated headers config.h / x264_config.h (no hand-written stand-ins for generated code);
#   2. gcc directly on the handful of library sources (list below == SRCS/SRCS_X of the reference
#      Makefile:19-31,66) with the same CFLAGS the reference configures, once per bit depth;
#   3. oracle/ref_harness.c (our own file) which #includes encoder/analyse.c of the reference so the
#      static slicetype/me functions are reachable, linked into libx264ref{8,10}.so.
#
#   4. the same 8-bit sources once more under a configuration whose only difference is HAVE_OPENCL (configure run without
#      --disable-opencl: "#define HAVE_OPENCL (BIT_DEPTH==8)", configure:1479-1491), with OUR translation unit
#      x264_amd/csrc/slicetype_hip.c standing in for the two files the reference adds in that build (encoder/slicetype-cl.c,
#      common/opencl.c; Makefile:254): libx264ref8hip.so is the unmodified reference encoder whose accelerator hook
#      (encoder/slicetype.c:878-897) calls libx264hip.so.  No OpenCL code, header stand-in or generated file is involved:
#      the reference's own extras/cl.h supplies the types its structs name.
#
# Products: oracle/_ref/libx264ref8.so, oracle/_ref/libx264ref10.so  (C-ABI, used via ctypes by
# tests/ and by bench.py's cpu_baseline leg only), oracle/_ref/libx264ref8hip.so (tests only).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${X264_REF:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -f "$REF/encoder/slicetype.c" ]; then
    echo "build_ref: reference tree not found at $REF (expected on the GPU box); keeping prebuilt files" >&2
    exit 0
fi
mkdir -p "$OUT/cfg" "$OUT/obj"
if [ ! -f "$OUT/cfg/config.h" ]; then
    (cd "$OUT/cfg" && "$REF/configure" --disable-asm --disable-opencl --disable-avs --disable-swscale \
        --disable-lavf --disable-ffms --disable-gpac --disable-lsmash --enable-static --enable-pic \
        --disable-cli > configure.log 2>&1) || { cat "$OUT/cfg/configure.log"; exit 1; }
fi
# CFLAGS exactly as the reference configures them (config.mak), re-rooted to our include dirs.
CFLAGS="-Wno-maybe-uninitialized -O3 -ffast-math -m64 -w -I$OUT/cfg -I$REF -std=gnu99 -D_GNU_SOURCE \
 -mpreferred-stack-boundary=6 -fPIC -fomit-frame-pointer -fno-tree-vectorize -fvisibility=hidden"
SRCS="common/osdep.c common/base.c common/cpu.c common/tables.c"
# encoder/analyse.c is deliberately absent: ref_harness.c includes it as a translation unit.
SRCS_X="common/mc.c common/predict.c common/pixel.c common/macroblock.c common/frame.c common/dct.c \
 common/cabac.c common/common.c common/rectangle.c common/set.c common/quant.c common/deblock.c \
 common/vlc.c common/mvpred.c common/bitstream.c encoder/me.c encoder/ratecontrol.c encoder/set.c \
 encoder/macroblock.c encoder/cabac.c encoder/cavlc.c encoder/encoder.c encoder/lookahead.c \
 common/threadpool.c"
pids=()
compile() { # src obj extra-flags
    if [ ! -f "$2" ] || [ "$REF/$1" -nt "$2" ]; then
        gcc $CFLAGS $3 -c "$REF/$1" -o "$2" &
        pids+=($!)
        if [ ${#pids[@]} -ge 8 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
    fi
}
COMMON_OBJS=""
for s in $SRCS; do
    o="$OUT/obj/$(echo "$s" | tr '/' '_' | sed 's/\.c$//').o"
    compile "$s" "$o" ""
    COMMON_OBJS="$COMMON_OBJS $o"
done
for depth in 8 10; do
    hbd=$([ $depth = 8 ] && echo 0 || echo 1)
    for s in $SRCS_X; do
        o="$OUT/obj/$(echo "$s" | tr '/' '_' | sed 's/\.c$//')-$depth.o"
        compile "$s" "$o" "-DHIGH_BIT_DEPTH=$hbd -DBIT_DEPTH=$depth"
    done
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
for depth in 8 10; do
    hbd=$([ $depth = 8 ] && echo 0 || echo 1)
    OBJS=""
    for s in $SRCS_X; do OBJS="$OBJS $OUT/obj/$(echo "$s" | tr '/' '_' | sed 's/\.c$//')-$depth.o"; done
    gcc $CFLAGS -DHIGH_BIT_DEPTH=$hbd -DBIT_DEPTH=$depth -I"$HERE" -Werror=implicit-function-declaration -c "$HERE/ref_harness.c" -o "$OUT/obj/ref_harness-$depth.o"
    gcc -shared -o "$OUT/libx264ref$depth.so" "$OUT/obj/ref_harness-$depth.o" $OBJS $COMMON_OBJS -lm -lpthread
done
# ---- 4. the reference with its accelerator hook bound to libx264hip.so
mkdir -p "$OUT/cfg_hip"
if [ ! -f "$OUT/cfg_hip/config.h" ]; then
    (cd "$OUT/cfg_hip" && "$REF/configure" --disable-asm --disable-avs --disable-swscale \
        --disable-lavf --disable-ffms --disable-gpac --disable-lsmash --enable-static --enable-pic \
        --disable-cli > configure.log 2>&1) || { cat "$OUT/cfg_hip/configure.log"; exit 1; }
fi
grep -q 'define HAVE_OPENCL (BIT_DEPTH==8)' "$OUT/cfg_hip/config.h" || { echo "build_ref: configure did not enable the accelerator seam" >&2; exit 1; }
CFLAGS_HIP="${CFLAGS/-I$OUT\/cfg /-I$OUT/cfg_hip }"
pids=()
CFLAGS_SAVE="$CFLAGS"; CFLAGS="$CFLAGS_HIP"
OBJS=""
for s in $SRCS_X; do
    o="$OUT/obj/$(echo "$s" | tr '/' '_' | sed 's/\.c$//')-8hip.o"
    compile "$s" "$o" "-DHIGH_BIT_DEPTH=0 -DBIT_DEPTH=8"
    OBJS="$OBJS $o"
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
REPO="$(dirname "$HERE")"
gcc $CFLAGS -DHIGH_BIT_DEPTH=0 -DBIT_DEPTH=8 -I"$REPO/include" -Wall -Werror=implicit-function-declaration -c "$REPO/x264_amd/csrc/slicetype_hip.c" -o "$OUT/obj/slicetype_hip-8.o"
gcc $CFLAGS -DHIGH_BIT_DEPTH=0 -DBIT_DEPTH=8 -I"$HERE" -Werror=implicit-function-declaration -c "$HERE/ref_harness.c" -o "$OUT/obj/ref_harness-8hip.o"
gcc -shared -o "$OUT/libx264ref8hip.so" "$OUT/obj/ref_harness-8hip.o" "$OUT/obj/slicetype_hip-8.o" $OBJS $COMMON_OBJS -lm -lpthread -ldl
CFLAGS="$CFLAGS_SAVE"
echo "build_ref: built $OUT/libx264ref8.so $OUT/libx264ref10.so $OUT/libx264ref8hip.so"
