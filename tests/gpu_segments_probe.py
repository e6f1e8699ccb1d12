"""Perf probe for frames cut into several launches: n full-screen unit-scale layers on either side of one scaled box,
timed as one general launch, under the planner's own policy, and cut at every class change (run through gpurun)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import kops as K
import scenes as S
import server_amd as A
from bench import Channel, run_steps


def mixed(W, H, bits, n, box):
    ops = [S.layer(S.ycbcr(W, H, "422", "video", 500 + i, bits=bits), opacity=0.9) for i in range(n)]
    ops.append(S.layer(S.bgra(W // 2, H // 2, "premul", 540, bits=bits), fill_translation=[0.0, 0.0] if box == 1.0 else [0.55, 0.5],
                       fill_scale=[box * 0.999, box * 0.999]))
    ops += [S.layer(S.ycbcr(W, H, "422", "video", 560 + i, bits=bits), opacity=0.8) for i in range(n)]
    return S.scene("mixed", W, H, ops, bits=bits)


L = A.lib()
for (W, H, bits) in ((3840, 2160, 16), (1920, 1080, 8)):
    for box in (0.3, 1.0):
        for n in (1, 2, 3, 4, 8):
            sc = mixed(W, H, bits, n, box)
            row = []
            # "policy": today the box is sampled by the 4-pixels-per-lane kernel itself (one launch); "cut": the round-1
            # plan, the box in a launch of the general kernel of its own between two launches of the fast one
            for name, policy in (("general", None), ("policy", (1.0, 8)), ("cut", (1.0, 8, False))):
                if policy is None:
                    L.cghip_debug_force_general_kernel(1)
                else:
                    K.set_cut_policy(*policy)
                ch = Channel(A, S, sc, 0, 1)
                run_steps(ch, 60)
                before = ch.mixer.device_info()["kernel_launches"]
                ms = run_steps(ch, 40)
                launches = (ch.mixer.device_info()["kernel_launches"] - before) // 40
                row.append("%s %.3f ms (%d)" % (name, float(np.mean(ms)), launches))
                ch.close()
                L.cghip_debug_force_general_kernel(0)
                K.set_cut_policy()
            print("%dx%d %2d-bit box %.1f  %d+1+%d layers:  %s" % (W, H, bits, box, n, n, "   ".join(row)), flush=True)
