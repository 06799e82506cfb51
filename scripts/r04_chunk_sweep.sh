cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04i
for c in 32 64 128 256; do
  echo "== chunk $c" | tee -a gpurun_out/r04i/summary.txt
  X264HIP_LA_CHUNK=$c timeout 300 python scripts/window_profile.py --modes plain --passes 2 2>&1 | grep '^{' | cut -c1-330 | tee -a gpurun_out/r04i/summary.txt
done
for c in 64 256; do
  echo "== headline chunk $c" | tee -a gpurun_out/r04i/summary.txt
  X264HIP_LA_CHUNK=$c timeout 300 python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check 2>&1 | grep '^{' | cut -c1-120 | tee -a gpurun_out/r04i/summary.txt
  X264HIP_LA_CHUNK=$c timeout 300 python bench.py --no-cpu-baseline --no-primitives --no-extra --no-check --inflight 1 2>&1 | grep '^{' | cut -c1-120 | tee -a gpurun_out/r04i/summary.txt
done
