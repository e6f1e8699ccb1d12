#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcp
PROBE_BITS=16 timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmcp -o p -- python3 tests/gpu_probe.py > gpurun_out/probe.log 2> gpurun_out/probe.err || { echo failed; tail -3 gpurun_out/probe.err; exit 1; }
grep -v amdgpu gpurun_out/probe.log
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmcp/p_counter_collection.csv')))
per=collections.OrderedDict()
for r in rows:
    if 'cghip' not in r['Kernel_Name']: continue
    per.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(per)
# 7 dispatches per scene (2 warm + 5)
for g in range(0,len(ids),7):
    d=per[ids[g+6]] if g+6<len(ids) else per[ids[-1]]
    print({k: round(v/1e6,1) for k,v in d.items()})
PY
