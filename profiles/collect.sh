#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against (run through gpurun):
#   1. --kernel-trace --stats of the default bench command  -> profiles/<tag>_kernel_stats.csv
#   2. HBM traffic counters in their own --pmc pass          -> profiles/<tag>_pmc_traffic.csv + pmc_traffic.json
# Usage: bash profiles/collect.sh r01
set -o pipefail
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end > $OUT/bench_stats.json 2> $OUT/stats.err || { echo "stats run failed"; tail -3 $OUT/stats.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/fetch.err || { echo "fetch pass failed"; tail -3 $OUT/fetch.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/write.err || { echo "write pass failed"; tail -3 $OUT/write.err; exit 1; }
ls -R $OUT | head -30
