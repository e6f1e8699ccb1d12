#!/bin/bash
# Same-box comparison of several builds (run through gpurun): every directory under server_amd/csrc/build/ab/ named in
# $1 (comma list; "tree" = the libraries in the tree) runs the probe stacks $2 at bit depths $3.
# Usage: bash tests/gpu_abn.sh base,v1,tree plain,chroma 16,8
set -o pipefail
DIRS=${1:-base,tree}
KINDS=${2:-plain,key,chroma,levels,bgra}
BITS=${3:-16,8}
mkdir -p gpurun_out
for side in ${DIRS//,/ }; do
  if [ $side = tree ]; then unset CGHIP_LIB_DIR; else export CGHIP_LIB_DIR=$PWD/server_amd/csrc/build/ab/$side; fi
  echo "== $side"
  if [ -z "$ABN_NO_CONFIGS" ]; then
    timeout -k 10 300 python -u tests/gpu_config_times.py > gpurun_out/abn_config_$side.log 2>&1 || { echo "config times failed ($side)"; tail -5 gpurun_out/abn_config_$side.log; exit 1; }
    grep -v amdgpu.ids gpurun_out/abn_config_$side.log | cut -c1-60
  fi
  PROBE_BITS=$BITS PROBE=$KINDS PROBE_LAYERS=16 PROBE_SETTLE=100 timeout -k 10 400 python -u tests/gpu_probe.py > gpurun_out/abn_probe_$side.log 2>&1 || { echo "probe failed ($side)"; tail -5 gpurun_out/abn_probe_$side.log; exit 1; }
  grep -v amdgpu.ids gpurun_out/abn_probe_$side.log
done
