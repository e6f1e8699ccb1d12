#!/bin/bash
# Counter pass over tests/gpu_probe.py stacks (run through gpurun): per kernel, counters per launch.
#   bash profiles/probe_counters.sh g_rect [general]
set -o pipefail
KINDS=${1:-g_rect}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PROBE_LAYERS=16 PROBE_BITS=${PROBE_BITS:-8} PROBE=$KINDS PROBE_SETTLE=3
[ "$2" = general ] && export PROBE_FORCE_GENERAL=1
OUT=gpurun_out/probe_pmc_${KINDS//,/_}${2:+_$2}
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT -o v -- python3 tests/gpu_probe.py > $OUT/probe.txt 2> $OUT/err.txt || { echo "pass failed"; tail -3 $OUT/err.txt; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1
for k in acc:
    print(k, "launches", n[k], {c: round(v / n[k] / 1e6, 2) for c, v in acc[k].items()}, "(M per launch)")
PY
cat $OUT/probe.txt
