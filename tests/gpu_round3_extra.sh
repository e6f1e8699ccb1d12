#!/bin/bash
# round 3: readback on its own stream, config 5 / config 4 in the product's topology (run through gpurun)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread -k "route or routed or config5 or two_channels or churn" > gpurun_out/gpu_route_tests.log 2>&1 || { echo "tests failed"; tail -30 gpurun_out/gpu_route_tests.log; exit 1; }
tail -2 gpurun_out/gpu_route_tests.log
for wl in config3_8bit config3_16bit config4; do
  timeout -k 10 300 python bench.py --workload $wl --readback --no-cpu-baseline --no-end-to-end > gpurun_out/bench_rb_$wl.json 2> gpurun_out/bench_rb_$wl.err || { echo "bench $wl failed"; tail -5 gpurun_out/bench_rb_$wl.err; exit 1; }
done
timeout -k 10 300 python bench.py --workload config5 --gpus 1 --steps 100 > gpurun_out/bench_config5_1gpu.json 2> gpurun_out/bench_config5_1gpu.err || { echo "config5 failed"; tail -5 gpurun_out/bench_config5_1gpu.err; exit 1; }
timeout -k 10 300 python bench.py --workload config5 --gpus 2 --stand-in --steps 100 > gpurun_out/bench_config5_2standin.json 2> gpurun_out/bench_config5_2standin.err || { echo "config5 stand-in failed"; tail -5 gpurun_out/bench_config5_2standin.err; exit 1; }
timeout -k 10 300 python bench.py --workload config4_native --gpus 1 --steps 500 > gpurun_out/bench_config4_native.json 2> gpurun_out/bench_config4_native.err || { echo "config4 native failed"; tail -5 gpurun_out/bench_config4_native.err; exit 1; }
python3 - <<'PY'
import json
for wl in ("config3_8bit", "config3_16bit", "config4"):
    d = json.load(open("gpurun_out/bench_rb_%s.json" % wl))
    rb = d["readback"]
    print("%-14s resident %.0f fps (kernel %.4f ms) | with readback %.0f fps, %.4f ms/step, kernel %.4f ms, D2H %.1f GB/s" % (
        wl, d["value"], d["roofline"]["avg_kernel_ms"], rb["frames_per_s"], rb["ms_per_step"], rb["avg_kernel_ms"], rb["d2h_GBps"]))
for n in ("config5_1gpu", "config5_2standin", "config4_native"):
    d = json.load(open("gpurun_out/bench_%s.json" % n))
    print("%-18s %.0f fps aggregate, %d channels, %d GPU(s), peer %.2f GB/s (%d copies) %s" % (
        n, d["value"], d["config"]["channels"], d["n_gpus"], d["peer_copy_GBps"], d["peer_copies"], d["peer_access"]))
PY
