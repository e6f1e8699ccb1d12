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
L.cghip_debug_fast_rows.argtypes = [__import__("ctypes").c_int]
run_steps(ch, 200)  # let the clocks settle
for rep in range(3):
    for rows, kb in ((4, 0), (1, 0), (2, 0), (4, 32), (4, 40), (4, 0)):
        L.cghip_debug_fast_rows(rows)
        L.cghip_debug_pad_lds(kb * 1024)
        run_steps(ch, 5)
        ms = run_steps(ch, 60)
        print("rep %d rows/wg %d  LDS pad %2d KB: kernel %.3f ms (min %.3f)" % (rep, rows, kb, np.mean(ms), np.min(ms)), flush=True)
L.cghip_debug_fast_rows(4)
L.cghip_debug_pad_lds(0)
ch.close()
