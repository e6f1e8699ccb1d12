#!/bin/bash
# Direct readback (the launch stores into the pinned buffer) against the D2H copy on the readback stream (run through gpurun)
set -o pipefail
for d in 0 1; do
  for b in 8 16; do
    echo "CGHIP_READBACK_DIRECT=$d bits $b: $(CGHIP_READBACK_DIRECT=$d RB_BITS=$b timeout -k 10 200 python -u tests/gpu_readback_trace.py 2>&1 | grep 'ms per frame')"
  done
done
CGHIP_READBACK_DIRECT=1 timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3
