#!/bin/bash
# full check after a build change: every GPU test, config times, the per-kind probes of both kernels (through gpurun)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { echo "tests failed"; tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 200 python -u tests/gpu_config_times.py > gpurun_out/config_times.log 2>&1 || { echo "config times failed"; tail -5 gpurun_out/config_times.log; exit 1; }
grep -v amdgpu.ids gpurun_out/config_times.log
PROBE=plain,key,chroma,levels,bgra,g_blend PROBE_LAYERS=16 PROBE_SETTLE=100 PROBE_BITS=16 timeout -k 10 300 python -u tests/gpu_probe.py > gpurun_out/probe_kinds.log 2>&1 || { echo "probe failed"; tail -5 gpurun_out/probe_kinds.log; exit 1; }
grep -v amdgpu.ids gpurun_out/probe_kinds.log
PROBE=g_rect,g_rot,g_pers,g_bgra_rect,g_unit PROBE_LAYERS=16 PROBE_SETTLE=100 timeout -k 10 300 python -u tests/gpu_probe.py > gpurun_out/probe_general.log 2>&1 || { echo "probe failed"; tail -5 gpurun_out/probe_general.log; exit 1; }
grep -v amdgpu.ids gpurun_out/probe_general.log
