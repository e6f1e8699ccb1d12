"""One-off long fuzz, 16-bit frames (run through gpurun; not collected by pytest): the planner's choice against the
general kernel forced on many box / fast-class scenes. The two kernels fuse different multiply-adds at 16 bits, so
the frames may differ by a code; anything beyond +-2 LSB16 at more than isolated amplifying pixels would be a bug in
the sampled 16-bit bodies.   python tests/gpu_long_fuzz16.py [first_seed] [last_seed]"""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes as S
import server_amd as A
from fuzz_scenes import fuzz_fast_scene, fuzz_box_scene

lo, hi = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 4000
m = A.Mixer(channel_id=9, gpu=0)
L = A.lib()
hist = collections.Counter()
worst = []
for seed in range(lo, hi):
    for sc in (fuzz_box_scene(seed, 16, seed % 2 == 1), fuzz_fast_scene(seed, 16, seed % 2 == 0)):
        cache = {}
        got = S.run_product(sc, mixer=m, planes_cache=cache)
        L.cghip_debug_force_general_kernel(1)
        general = S.run_product(sc, mixer=m, planes_cache=cache)
        L.cghip_debug_force_general_kernel(0)
        st = S.diff_stats(got, general)
        hist[min(st["max"], 9)] += 1
        if st["gt2"] > 1e-3:
            worst.append((sc["name"], st))
    if seed % 500 == 0:
        print("seed", seed, dict(sorted(hist.items())), flush=True)
m.close()
print("max |fast - general| per scene (9 = 9 or more):", dict(sorted(hist.items())))
print("scenes with more than 1e-3 of samples beyond +-2:", len(worst), worst[:10])
sys.exit(1 if worst else 0)
