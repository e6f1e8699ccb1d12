"""Worker for tests/test_dist_gloo.py: one rank of a world_size-N gloo job exercising bench.py's N>1 plumbing on CPU
(channel -> rank sharding, barrier, max-over-ranks timing, whole-job value)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CGHIP_DIST_BACKEND"] = "gloo"

import bench  # noqa: E402


def main():
    out_path = sys.argv[1]
    steps = 7
    rank, world, local_rank, barrier, allreduce_max = bench.dist_setup(world_hint := int(os.environ["WORLD_SIZE"]))
    assert world == world_hint and 0 <= rank < world
    barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))  # rank r "renders" for 50(r+1) ms: the job time is the slowest rank's
    mine = time.perf_counter() - t0
    barrier()
    dt = allreduce_max(mine)
    assert dt >= mine - 1e-9
    channels = bench.shard_channels(10, world)  # 10 channels over the ranks
    rec = {"rank": rank, "world": world, "mine": mine, "max": dt, "channels": channels[rank],
           "value": bench.whole_job_value(world, steps, dt)}
    with open("%s.%d" % (out_path, rank), "w") as f:
        json.dump(rec, f)
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
