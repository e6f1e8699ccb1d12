"""One-off bug hunt: many more random scenes than tests/test_fuzz.py runs. usage: gpu_fuzz_hunt.py FIRST COUNT"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes as S
import server_amd as A
from fuzz_scenes import fuzz_scene, fuzz_fast_scene

first, count = int(sys.argv[1]), int(sys.argv[2])
m = A.Mixer(channel_id=3, gpu=0)
bad = 0
for seed in range(first, first + count):
    for gen in (fuzz_scene, fuzz_fast_scene):
        sc = gen(seed, 8)
        cache = {}
        try:
            got = S.run_product(sc, mixer=m, planes_cache=cache)
        except Exception as e:
            print("seed", seed, gen.__name__, "EXCEPTION", str(e)[:200], flush=True)
            bad += 1
            continue
        ref = S.run_oracle(sc, "cpu", cache)
        st = S.diff_stats(got, ref)
        pow_used = any(o[0] == "push" and o[1].get("levels_gamma", 1.0) != 1.0 for o in sc["ops"])
        if st["max"] > (1 if pow_used else 0) or st["gt0"] > 1e-3:
            print("seed", seed, gen.__name__, st, "pow" if pow_used else "", flush=True)
            bad += 1
    if seed % 500 == 0:
        print("... seed", seed, flush=True)
print("done: %d scenes x2, %d flagged" % (count, bad))
m.close()
