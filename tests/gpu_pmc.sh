#!/bin/bash
# usage: bash tests/gpu_pmc.sh "<counters>" [tag]   -- one PMC pass over a short bench run (through gpurun)
set -o pipefail
TAG=${2:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_$TAG
timeout -s KILL 120 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG -o p -- python3 bench.py --settle 0.5 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end $BENCH_ARGS > gpurun_out/pmc_$TAG.json 2> gpurun_out/pmc_$TAG.err || { echo pmc failed; tail -3 gpurun_out/pmc_$TAG.err; exit 1; }
python3 - $TAG <<'PY'
import csv, collections, sys
agg=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmc_%s/p_counter_collection.csv' % sys.argv[1])):
    if 'cghip' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("PMC", {k: round(sum(v)/len(v)/1e6,3) for k,v in agg.items()})
PY
