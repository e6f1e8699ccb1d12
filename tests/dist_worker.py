"""Worker for tests/test_dist_gloo.py: one rank of a world_size-N gloo job exercising bench.py's N>1 path (channel ->
rank sharding, barrier, max-over-ranks timing, whole-job value) AROUND THE MIXER: every rank drives the channels of its
shard through push / visit / pop / render. With a GPU visible (ranks share GPU 0 on a one-GPU box) the frames are
composited and checked against the oracle; without one the mixer is the dry-run one (host logic only: transform stack,
layer tree, draw list, planner)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CGHIP_DIST_BACKEND"] = "gloo"

import bench  # noqa: E402


def main():
    out_path = sys.argv[1]
    steps = 7
    rank, world, local_rank, barrier, allreduce_max = bench.dist_setup(world_hint := int(os.environ["WORLD_SIZE"]))
    assert world == world_hint and 0 <= rank < world
    import numpy as np
    import scenes as S
    import server_amd as A
    import kops as K
    channels = bench.shard_channels(10, world)  # 10 channels over the ranks
    on_gpu = os.environ.get("CGHIP_DIST_GPU") == "1" and A.device_count() > 0
    W, H = 160, 90
    scenes = {c: S.config4(W, H, delta=0.1 * c) for c in channels[rank]}  # every channel at another point of its MIX
    mixers = {c: A.Mixer(channel_id=c, gpu=0 if on_gpu else -2) for c in channels[rank]}
    want = {c: S.run_oracle(scenes[c], "cpu") for c in channels[rank]} if on_gpu else {}
    barrier()
    t0 = time.perf_counter()
    rendered = 0
    for _ in range(steps):
        for c in channels[rank]:
            if on_gpu:
                got = S.run_product(scenes[c], mixer=mixers[c])
                assert np.array_equal(got, want[c]), (rank, c)
            else:  # dry run: the draw list the renderer would issue, through the planner
                S.feed_product(mixers[c], scenes[c], via_factory=False)
                draws, surfaces = mixers[c].flatten(W, H)
                ops, _full = K.plan(A.Format(W, H, W, H, 8, 0), surfaces, draws)
                assert len(draws) == 4 and len(ops) >= 2  # background, two MIX items, the composite (one op with its item when a MIX weight is 0)
                try:  # a dry-run mixer cannot render; the attempt ends the frame (the layer tree is consumed)
                    mixers[c].render(W, H)
                except A.CGHIPError:
                    pass
            rendered += 1
    time.sleep(0.05 * (rank + 1))  # rank r takes 50(r+1) ms longer: the job time is the slowest rank's
    mine = time.perf_counter() - t0
    barrier()
    dt = allreduce_max(mine)
    assert dt >= mine - 1e-9
    for m in mixers.values():
        m.close()
    rec = {"rank": rank, "world": world, "mine": mine, "max": dt, "channels": channels[rank], "rendered": rendered,
           "mode": "gpu" if on_gpu else "dry", "value": bench.whole_job_value(world, steps, dt)}
    with open("%s.%d" % (out_path, rank), "w") as f:
        json.dump(rec, f)
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
