"""Where the PCIe-inclusive frame time goes (config 3, through gpurun)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes as S
import server_amd as A

sc = S.config3(bits=16)
m = A.Mixer(channel_id=5, gpu=0)
cache = {}
S.run_product(sc, mixer=m, planes_cache=cache)
W, H = sc["width"], sc["height"]
for it in range(4):
    t0 = time.perf_counter()
    S.feed_product(m, sc, cache)
    t1 = time.perf_counter()
    p = m.render_async(W, H, bits=16)
    t2 = time.perf_counter()
    out = m.get(p)
    t3 = time.perf_counter()
    print("feed %.1f ms  render_async %.1f ms  get %.1f ms  | info %s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
          {k: m.device_info()[k] for k in ("uploads", "planes_live")}), flush=True)
m.close()
