"""How the headline launch's duration evolves in the FIRST process on a fresh box (run through gpurun): mean kernel
time per half second of continuous load, with the shader clock if rocm-smi lets an ordinary user read it."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes as S
import server_amd as A
from bench import Channel, run_steps


def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
        keep = [l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "Temperature (Sensor junction)" in l]
        return " | ".join(k.split("GPU[0]")[-1].strip(" :\t") for k in keep)[:160]
    except Exception as e:
        return "rocm-smi: %s" % e


ch = Channel(A, S, S.config3(bits=16), 0, 1)
total = float(os.environ.get("PROBE_SECONDS", "16"))
t0 = time.perf_counter()
nxt = 0.5
acc = []
print("idle:", sclk(), flush=True)
while time.perf_counter() - t0 < total:
    acc += run_steps(ch, 20)
    if time.perf_counter() - t0 >= nxt:
        smi = sclk() if int(nxt * 2) % 4 == 0 else ""
        print("t=%5.1f s  kernel %.4f ms (min %.4f max %.4f, %d launches)  %s" % (nxt, np.mean(acc), min(acc), max(acc), len(acc), smi), flush=True)
        acc = []
        nxt += 0.5
ch.close()
