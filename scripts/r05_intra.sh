#!/bin/bash
# round 5, GPU call: intra_kernel with its mode loop unrolled (default) against the loop form (-DINTRA_UNROLL=0, libx264hip_nounroll.so)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05intra; mkdir -p $out
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x ) > $out/parity.log 2>&1; echo "parity rc=$?"; tail -2 $out/parity.log
short="--no-cpu-baseline --no-primitives --no-extra"
for rep in 1 2; do
for V in "" "_nounroll"; do
  X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/libx264hip$V.so timeout 300 python bench.py $short > $out/b$V.log 2>&1
  python - "lib$V" $out/b$V.log <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    k=j.get("roofline_kernels",{})
    print(sys.argv[1], "fps %.0f" % j["value"], json.dumps(k)[:900])
except Exception as e: print(sys.argv[1],"FAILED",e)
PY
done
done
