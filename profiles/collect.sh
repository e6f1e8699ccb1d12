#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against (run through gpurun):
#   1. --kernel-trace --stats of the default bench command  -> profiles/<tag>_kernel_stats.csv
#   2. HBM traffic counters, one --pmc pass each (FETCH_SIZE, WRITE_SIZE) -> profiles/<tag>_pmc_traffic.json + pmc_traffic.json
#   3. VALU / wave counters in a third --pmc pass            -> profiles/<tag>_pmc_valu.json
# Usage: bash profiles/collect.sh r01
set -o pipefail
TAG=${1:-r01}
WL=${2:-config3_16bit}   # bench.py --workload; tag the output accordingly, e.g. collect.sh r02_config2 config2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --workload $WL --no-cpu-baseline --no-end-to-end > $OUT/bench_stats.json 2> $OUT/stats.err || { echo "stats run failed"; tail -3 $OUT/stats.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 bench.py --workload $WL --settle 0.5 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/fetch.err || { echo "fetch pass failed"; tail -3 $OUT/fetch.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 bench.py --workload $WL --settle 0.5 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/write.err || { echo "write pass failed"; tail -3 $OUT/write.err; exit 1; }
# 3. the vector-ALU side (the compositor is VALU-issue bound): instruction counts, wave cycles, clock
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/valu -o v -- python3 bench.py --workload $WL --settle 0.5 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/valu.err || { echo "valu pass failed"; tail -3 $OUT/valu.err; exit 1; }
python3 profiles/summarize.py $TAG
