#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_lookahead.py tests/test_gpu_fuzz.py tests/test_gpu_host_fed.py -x -q -m gpu 2>&1 | tail -3 | tee $out/tests.txt
run() { # label lib env...
  label=$1; lib=$2; shift; shift
  v=$(env "$@" X264HIP_LIB=$GRAFT_REPO_ROOT/x264_amd/$lib python bench.py --no-cpu-baseline --no-primitives --no-extra --steps 20 --warmup 3 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['us_per_search'], j['paced_fps'])")
  echo "$label $v" | tee -a $out/ab.txt
}
for i in 1 2; do
run prev libx264hip_prev.so A=0
run new_wps3 libx264hip.so A=0
run new_wps0 libx264hip.so X264HIP_ROWS_WAVES=0
run new_wps2 libx264hip.so X264HIP_ROWS_WAVES=2
run new_wps5 libx264hip.so X264HIP_ROWS_WAVES=5
done
