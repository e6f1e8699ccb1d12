#!/bin/bash
# Several channels on one GPU, every frame read back: D2H copy on the readback stream against direct readback (run through gpurun)
set -o pipefail
for d in 0 1; do
  for wl in "config5 --channels 8 --frames 60" "config4 --channels 32 --frames 300" "config4 --channels 4 --frames 300"; do
    echo "CGHIP_READBACK_DIRECT=$d $wl: $(CGHIP_READBACK_DIRECT=$d timeout -k 10 200 tests/native/route_driver --workload $wl --gpus 1 --readback | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.0f frames/s, %.0f per channel" % (d["frames_per_s"], d["per_channel_fps"]))')"
  done
done
