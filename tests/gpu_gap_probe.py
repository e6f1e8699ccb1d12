"""Perf probe: wall-clock time per frame of one channel with and without the hipEvent pair around the launches
(CGHIP_RENDER_UNTIMED), i.e. what the timing events cost on the compositor's queue (run through gpurun)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenes as S
import server_amd as A
from bench import Channel, run_steps, WORKLOADS

name = os.environ.get("PROBE_WORKLOAD", "config3_16bit")
sc = WORKLOADS[name](S)
ch = Channel(A, S, sc, 0, 1)
ch.mixer.H.cghost_mixer_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
run_steps(ch, 3000)  # clocks
for timed in (1, 0, 1, 0):
    ch.mixer.H.cghost_mixer_set_kernel_timing(ch.mixer.h, timed)
    run_steps(ch, 200)
    t0 = time.perf_counter()
    run_steps(ch, 1000)
    dt = time.perf_counter() - t0
    print("%s kernel timing %d: %.4f ms per frame" % (name, timed, dt), flush=True)  # dt is seconds per 1000 frames
ch.close()
