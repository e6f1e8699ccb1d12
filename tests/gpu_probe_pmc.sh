#!/bin/bash
# per-layer-type instruction counts: PROBE=... PROBE_BITS=... PROBE_SUB=... bash tests/gpu_probe_pmc.sh  (through gpurun)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmcp -o p -- python3 tests/gpu_probe.py > gpurun_out/probe.log 2> gpurun_out/probe.err || { echo failed; tail -3 gpurun_out/probe.err; exit 1; }
grep -v amdgpu gpurun_out/probe.log
python3 - <<'PY'
import csv, collections, os
rows=list(csv.DictReader(open('gpurun_out/pmcp/p_counter_collection.csv')))
per=collections.OrderedDict()
for r in rows:
    if 'cghip' not in r['Kernel_Name']: continue
    per.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(per)
W=int(os.environ.get("PROBE_W","3840")); H=int(os.environ.get("PROBE_H","2160")); NL=int(os.environ.get("PROBE_LAYERS","8"))
den=NL*W*H/64
for g in range(0,len(ids),7):  # 7 dispatches per scene (2 warm + 5)
    d=per[ids[min(g+6,len(ids)-1)]]
    print("per layer-px: VALU %.1f TRANS %.2f SALU %.1f VMEM %.2f" % (d['SQ_INSTS_VALU']/den, d.get('SQ_INSTS_VALU_TRANS_F32',0)/den, d['SQ_INSTS_SALU']/den, d['SQ_INSTS_VMEM_RD']/den))
PY
