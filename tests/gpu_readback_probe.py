"""Readback probe (run through gpurun): what the box's D2H path moves by itself for the frame sizes of the BASELINE
configs (torch pinned copies on a side stream, idle GPU), and what a channel with resident sources reaches with every
frame read back on its readback stream, 2 / 3 / 4 frames in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import scenes as S
import server_amd as A
from bench import Channel, run_steps

for mb in (8.2944, 33.1776, 66.3552):
    n = int(mb * 1e6)
    d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(5):
            h.copy_(d, non_blocking=True)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            h.copy_(d, non_blocking=True)
        st.synchronize()
        dt = (time.perf_counter() - t0) / 50
    print("D2H alone %6.1f MB: %.3f ms = %.1f GB/s" % (mb, dt * 1e3, n / dt / 1e9), flush=True)

for name, sc in (("config3_8bit", S.config3(bits=8)), ("config3_16bit", S.config3(bits=16))):
    ch = Channel(A, S, sc, 0, 1)
    run_steps(ch, 100)
    fb = sc["width"] * sc["height"] * 4 * (sc["bits"] // 8)
    for inflight in (2, 3, 4):
        run_steps(ch, 10, in_flight=inflight, readback=True)
        t0 = time.perf_counter()
        ms = run_steps(ch, 100, in_flight=inflight, readback=True)
        dt = (time.perf_counter() - t0) / 100
        print("%-14s %d in flight: %.3f ms/step = %.0f fps, D2H %.1f GB/s, kernel %.3f ms" % (
            name, inflight, dt * 1e3, 1 / dt, fb / dt / 1e9, sum(ms) / len(ms)), flush=True)
    ch.close()
