#!/bin/bash
# parity, kernel times and HBM traffic after a kernel change (run through gpurun)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { echo "tests failed"; tail -20 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 200 python -u tests/gpu_config_times.py > gpurun_out/config_times.log 2>&1 || { echo "config times failed"; tail -5 gpurun_out/config_times.log; exit 1; }
grep -v amdgpu.ids gpurun_out/config_times.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in config3_16bit config2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${wl}_$c
    timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${wl}_$c -o p -- python3 bench.py --settle 0.5 --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > /dev/null 2> gpurun_out/pmc_${wl}_$c.err || { echo "pmc $wl $c failed"; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_*_*_SIZE')):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'cghip' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(d, k, 'mean per launch (raw counter units)', sum(v) / len(v), 'n', len(v))
PY
