"""Readback under rocprofv3 --kernel-trace --memory-copy-trace (run through gpurun): config 3 at 8 bits, resident
sources, every frame read back, 2 in flight. The trace shows where the time between two D2H copies goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenes as S
import server_amd as A
from bench import Channel, run_steps

bits = int(os.environ.get("RB_BITS", "8"))
sc = S.config3(bits=bits)
ch = Channel(A, S, sc, 0, 1)
run_steps(ch, 50)
run_steps(ch, 10, in_flight=2, readback=True)
t0 = time.perf_counter()
run_steps(ch, 60, in_flight=2, readback=True)
print("%.3f ms per frame" % ((time.perf_counter() - t0) / 60 * 1e3))
ch.close()
