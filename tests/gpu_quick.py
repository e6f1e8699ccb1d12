"""Ad-hoc GPU shake-out: product vs oracle on every small scene (run by hand through gpurun)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes as S
import server_amd as A

print("devices", A.device_count(), flush=True)
m = A.Mixer(1, 0)
worst = {}
for sc in S.all_golden_scenes():
    ref = S.run_oracle(sc, "cpu")
    try:
        got = S.run_product(sc, mixer=m)
    except Exception as e:
        print("%-40s ERROR %s" % (sc["name"], e), flush=True)
        continue
    st = S.diff_stats(got, ref)
    print("%-40s %s" % (sc["name"], st), flush=True)
for name, mk in (("config1", S.config1), ("config2", S.config2)):
    sc = mk()
    cache = {}
    t0 = time.time(); ref = S.run_oracle(sc, "cpu", cache); t1 = time.time()
    got = S.run_product(sc, mixer=m, planes_cache=cache)
    print(name, S.diff_stats(got, ref), "oracle %.2fs kernel %.3f ms" % (t1 - t0, m.last_kernel_ms), flush=True)
