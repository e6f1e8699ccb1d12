#!/bin/bash
# Same-box A/B of two builds (run through gpurun): server_amd/csrc/build/ab/base/ (a copy of the libraries of the commit
# the change started from) against the libraries in the tree. Usage: bash tests/gpu_ab.sh [probe kinds] [bits]
set -o pipefail
KINDS=${1:-plain,key,chroma,levels,bgra}
BITS=${2:-16,8}
mkdir -p gpurun_out
for side in base new; do
  if [ $side = base ]; then export CGHIP_LIB_DIR=$PWD/server_amd/csrc/build/ab/base; else unset CGHIP_LIB_DIR; fi
  timeout -k 10 300 python -u tests/gpu_config_times.py > gpurun_out/ab_config_$side.log 2>&1 || { echo "config times failed ($side)"; tail -5 gpurun_out/ab_config_$side.log; exit 1; }
  echo "== $side"; grep -v amdgpu.ids gpurun_out/ab_config_$side.log | cut -c1-110
  PROBE_BITS=$BITS PROBE=$KINDS PROBE_LAYERS=16 PROBE_SETTLE=100 timeout -k 10 400 python -u tests/gpu_probe.py > gpurun_out/ab_probe_$side.log 2>&1 || { echo "probe failed ($side)"; tail -5 gpurun_out/ab_probe_$side.log; exit 1; }
  grep -v amdgpu.ids gpurun_out/ab_probe_$side.log
done
