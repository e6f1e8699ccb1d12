#!/bin/bash
# Round check on the GPU box (run through gpurun): kernel times of the BASELINE configs, the blend-mode probe,
# the default bench line, then the rocprofv3 evidence (profiles/collect.sh). Every GPU step has its own time limit
# and the chain stops at the first failure.
set -o pipefail
TAG=${1:-r01}
mkdir -p gpurun_out
timeout -k 10 200 python -u tests/gpu_config_times.py > gpurun_out/config_times.log 2>&1 || { echo "config times failed"; tail -5 gpurun_out/config_times.log; exit 1; }
cat gpurun_out/config_times.log
PROBE=g_blend PROBE_LAYERS=16 PROBE_SETTLE=100 timeout -k 10 120 python -u tests/gpu_probe.py > gpurun_out/probe_blend.log 2>&1 || { echo "blend probe failed"; tail -5 gpurun_out/probe_blend.log; exit 1; }
grep -v amdgpu.ids gpurun_out/probe_blend.log
PROBE_LAYERS=16 PROBE_SETTLE=100 PROBE_BITS=16 timeout -k 10 200 python -u tests/gpu_probe.py > gpurun_out/probe_kinds.log 2>&1 || { echo "kind probe failed"; tail -5 gpurun_out/probe_kinds.log; exit 1; }
grep -v amdgpu.ids gpurun_out/probe_kinds.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { echo "bench failed"; tail -5 gpurun_out/bench_default.err; exit 1; }
cat gpurun_out/bench_default.json
bash profiles/collect.sh $TAG > gpurun_out/collect.log 2>&1 || { echo "collect failed"; tail -20 gpurun_out/collect.log; exit 1; }
tail -20 gpurun_out/collect.log
mkdir -p gpurun_out/profiles_back && cp profiles/*${TAG}* profiles/pmc_traffic.json gpurun_out/profiles_back/
