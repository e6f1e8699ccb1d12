#!/bin/bash
# A channel with resident sources and every frame read back, with plain streams and with the runtime's stream priorities
# (compositor lowest, readback highest), beside the box's D2H rate alone (run through gpurun)
set -o pipefail
python -u - <<'PY'
import torch, time
for mb in (33.1776, 66.3552):
    n = int(mb * 1e6)
    d = torch.empty(n, dtype=torch.uint8, device="cuda:0"); h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(5): h.copy_(d, non_blocking=True)
        st.synchronize(); t0 = time.perf_counter()
        for _ in range(50): h.copy_(d, non_blocking=True)
        st.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("D2H alone %6.1f MB: %.3f ms = %.1f GB/s" % (mb, dt * 1e3, n / dt / 1e9), flush=True)
PY
for p in 0 1; do
  for b in 8 16; do
    echo "CGHIP_STREAM_PRIORITIES=$p bits $b: $(CGHIP_STREAM_PRIORITIES=$p RB_BITS=$b timeout -k 10 200 python -u tests/gpu_readback_trace.py 2>&1 | grep 'ms per frame')"
  done
done
