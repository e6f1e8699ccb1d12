#!/bin/bash
# A/B of the one-format kernel instances (cghip_composite_fast_m) against the kernel with all fetch bodies, same box,
# same process order (run through gpurun)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u tests/gpu_config_times.py > gpurun_out/config_times.log 2>&1 || { echo "config times failed"; tail -5 gpurun_out/config_times.log; exit 1; }
grep -v amdgpu.ids gpurun_out/config_times.log
for nm in "" 1; do
  PROBE_NO_MONO=$nm PROBE=plain,key,chroma,levels,bgra PROBE_LAYERS=16 PROBE_SETTLE=100 timeout -k 10 300 python -u tests/gpu_probe.py > gpurun_out/probe_mono_$nm.log 2>&1 || { echo "probe failed"; tail -5 gpurun_out/probe_mono_$nm.log; exit 1; }
  echo "PROBE_NO_MONO=$nm"; grep -v amdgpu.ids gpurun_out/probe_mono_$nm.log
done
