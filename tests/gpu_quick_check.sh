#!/bin/bash
# parity + the default bench line after a runtime / kernel change (run through gpurun)
set -o pipefail
mkdir -p gpurun_out
# the first `import torch` on a fresh box pages in for a minute or two, and the first test starts two more processes
# that do the same: say so every 30 s (every test has its own --timeout, a real hang still ends the run)
( while sleep 30; do echo "[quick check] running"; done ) &
HEART=$!
trap 'kill $HEART 2>/dev/null' EXIT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { echo "tests failed"; tail -20 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { echo "bench failed"; tail -5 gpurun_out/bench_default.err; exit 1; }
for wl in config1 config2 config4 config3_8bit; do
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-end-to-end > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err || { echo "bench $wl failed"; exit 1; }
done
python3 - <<'PY'
import json
for n in ("default", "config1", "config2", "config4", "config3_8bit"):
    d = json.load(open("gpurun_out/bench_%s.json" % n))
    print("%-12s value %.1f fps  ms_per_step %.4f  kernel %.4f ms  frac %.3f  e2e %s  cpu %s" % (
        n, d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d.get("end_to_end_fps"),
        (d.get("cpu_baseline") or {}).get("value")))
PY
