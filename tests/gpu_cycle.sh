#!/bin/bash
# one GPU iteration: parity sweep, bench, instruction counters (run through gpurun)
set -o pipefail
timeout -k 10 240 python -u tests/gpu_quick.py > gpurun_out/quick.log 2>&1 || { echo quick failed; tail -5 gpurun_out/quick.log; exit 1; }
grep -v "'max': 0," gpurun_out/quick.log | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-end-to-end --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err || { echo bench failed; tail -5 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print("BENCH value %.1f fps  kernel %.3f ms  achieved %.0f GB/s frac %.3f" % (d['value'], d['roofline']['avg_kernel_ms'], d['roofline']['achieved'], d['roofline']['frac']))
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc -o p -- python3 bench.py --settle 0.5 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> gpurun_out/pmc.err || { echo pmc failed; exit 1; }
python - <<'PY'
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmc/p_counter_collection.csv')):
    if 'cghip' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("PMC", {k: round(sum(v)/len(v)/1e6,2) for k,v in agg.items()})
PY
