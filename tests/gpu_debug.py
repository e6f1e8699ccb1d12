import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes as S
import server_amd as A
m = A.Mixer(1, 0)
sc = [s for s in S.small_scenes() if s["name"] == "opacity_half_ties"][0]
ref = S.run_oracle(sc, "cpu"); got = S.run_product(sc, mixer=m)
bgp = S.make_plane(96, 64, 4, "premul", 11); fgp = S.make_plane(96, 64, 4, "premul", 3)
ys, xs, cs = np.nonzero(ref != got)
for y, x, c in list(zip(ys, xs, cs))[:12]:
    print("px", y, x, "ch", c, "bg", bgp[y, x], "fg", fgp[y, x], "ref", ref[y, x], "got", got[y, x])
