"""Kernel time of the BASELINE configs under the planner's policy and as one launch (run through gpurun)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import kops as K
import scenes as S
import server_amd as A
from bench import Channel, run_steps

L = A.lib()


def no_mono(on):  # absent from builds older than round 3 (tests/gpu_ab.sh runs those too)
    if hasattr(L, "cghip_debug_no_mono_kernels"):
        L.cghip_debug_no_mono_kernels(on)


for name, sc in (("config1", S.config1()), ("config2", S.config2()), ("config4", S.config4()),
                 ("config3_8bit", S.config3(bits=8)), ("config3_16bit", S.config3(bits=16))):
    row = []
    for label, setup in (("policy", lambda: None), ("one launch", lambda: K.set_cut_policy(max_segments=1)),
                         ("all-bodies kernel", lambda: no_mono(1)),
                         ("full-frame composites", lambda: L.cghip_debug_full_frame_composites(1)),
                         ("general kernel", lambda: L.cghip_debug_force_general_kernel(1))):
        setup()
        ch = Channel(A, S, sc, 0, 1)
        run_steps(ch, 100)
        before = ch.mixer.device_info()["kernel_launches"]
        ms = run_steps(ch, 50)
        launches = (ch.mixer.device_info()["kernel_launches"] - before) // 50
        row.append("%s %.4f ms (%d)" % (label, float(np.mean(ms)), launches))
        ch.close()
        K.set_cut_policy()
        L.cghip_debug_full_frame_composites(0)
        no_mono(0)
        L.cghip_debug_force_general_kernel(0)
    print("%-14s %s" % (name, "   ".join(row)), flush=True)
