#!/bin/bash
# One counter pass over tests/gpu_probe.py stacks for one build (run through gpurun): per kernel, counters per launch.
#   bash profiles/probe_pmc.sh <tree|ab dir name> <kinds> <bits> <tag> COUNTER...
set -o pipefail
SIDE=$1; KINDS=$2; BITS=$3; TAG=$4; shift 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ $SIDE != tree ]; then export CGHIP_LIB_DIR=$PWD/server_amd/csrc/build/ab/$SIDE; fi
export PROBE_LAYERS=16 PROBE_BITS=$BITS PROBE=$KINDS PROBE_SETTLE=3
OUT=gpurun_out/pmc_${TAG}_${SIDE}
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -o v -- python3 tests/gpu_probe.py > $OUT/probe.txt 2> $OUT/err.txt || { echo "pass failed"; tail -3 $OUT/err.txt; exit 1; }
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
