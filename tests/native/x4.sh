set -e
export PROBE_LAYERS=16 PROBE=g_rect,g_bgra_rect,g_unit PROBE_SETTLE=20
PROBE_BITS=8,16 timeout -k 10 300 python tests/gpu_probe.py
