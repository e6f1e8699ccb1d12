"""Occupancy experiment: kernel time of the bench workload with the workgroups per CU capped by unused LDS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes as S
import server_amd as A
from bench import Channel, run_steps

L = A.lib()
L.cghip_debug_pad_lds.argtypes = [__import__("ctypes").c_int]
sc = S.config3(bits=16)
ch = Channel(A, S, sc, 0, 1)
for kb in (0, 32, 40, 53, 64, 80):
    L.cghip_debug_pad_lds(kb * 1024)
    run_steps(ch, 3)
    ms = run_steps(ch, 20)
    print("LDS pad %3d KB -> <= %d workgroups (waves per SIMD) per CU: kernel %.3f ms" % (kb, 160 // kb if kb else 8, np.mean(ms)), flush=True)
L.cghip_debug_pad_lds(0)
ch.close()
