"""One-off long fuzz (run through gpurun; not collected by pytest): many more seeds than tests/test_fuzz.py of the scene
families whose kernels are new in round 2 - sampled boxes, fused blend-mode layers, uyvy / byte formats - in 8-bit
frames, where everything must be bit-identical: planner's choice == general kernel forced == unfused plan == oracle.
    python tests/gpu_long_fuzz.py [first_seed] [last_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes as S
import server_amd as A
from fuzz_scenes import fuzz_scene, fuzz_fast_scene, fuzz_box_scene, fuzz_nested_scene, fuzz_mix_scene

lo, hi = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 2500
m = A.Mixer(channel_id=9, gpu=0)
L = A.lib()
bad = 0
t0 = time.time()
for seed in range(lo, hi):
    for sc in (fuzz_box_scene(seed, 8, seed % 2 == 1), fuzz_fast_scene(seed, 8, seed % 2 == 0), fuzz_scene(seed, 8),
               fuzz_box_scene(seed, 8, seed % 2 == 0, mirror=True), fuzz_nested_scene(seed, 8), fuzz_mix_scene(seed, 8),
               fuzz_mix_scene(seed, 8, True), fuzz_mix_scene(seed, 8, seed % 2 == 0, True)):  # the last five: round 3
        cache = {}
        got = S.run_product(sc, mixer=m, planes_cache=cache)
        L.cghip_debug_force_general_kernel(1)
        general = S.run_product(sc, mixer=m, planes_cache=cache)
        L.cghip_debug_force_general_kernel(0)
        L.cghip_debug_no_layer_fusion(1)
        L.cghip_debug_no_mix_fusion(1)
        unfused = S.run_product(sc, mixer=m, planes_cache=cache)
        L.cghip_debug_no_layer_fusion(0)
        L.cghip_debug_no_mix_fusion(0)
        ref = S.run_oracle(sc, "cpu", cache)
        pow_used = any(o[0] == "push" and o[1].get("levels_gamma", 1.0) != 1.0 for o in sc["ops"])
        ok = np.array_equal(got, general) and np.array_equal(got, unfused)
        st = S.diff_stats(got, ref)
        ok = ok and ((st["max"] <= 1 and st["gt0"] <= 1e-4) if pow_used else st["max"] == 0)
        if not ok:
            bad += 1
            print("MISMATCH", sc["name"], S.diff_stats(got, general), S.diff_stats(got, unfused), st, flush=True)
    if seed % 100 == 0:
        print("seed", seed, "bad", bad, "%.0f s" % (time.time() - t0), flush=True)
m.close()
print("seeds %d..%d: %d mismatching scenes" % (lo, hi, bad))
sys.exit(1 if bad else 0)
