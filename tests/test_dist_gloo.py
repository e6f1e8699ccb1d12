"""CPU suite: the N>1 path of bench.py (one process per GPU, no data-path collective) rehearsed with world_size 2
on gloo. Channels are independent, so the only distributed pieces are the channel->rank shard, the
barrier and the max-over-ranks reduction of the timed region."""
import glob
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


import pytest


def test_world_size_2_gloo(tmp_path):
    _run_world_of_2(tmp_path, gpu=False)


@pytest.mark.gpu
def test_world_size_2_gloo_composites_on_the_gpu(tmp_path):
    """The same job with the ranks' channels compositing on the GPU (two processes share GPU 0 on a one-GPU box) and
    every frame checked against the oracle."""
    _run_world_of_2(tmp_path, gpu=True)


def _run_world_of_2(tmp_path, gpu):
    out = str(tmp_path / "rank")
    env = dict(os.environ, CGHIP_DIST_BACKEND="gloo", CGHIP_DIST_GPU="1" if gpu else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = sorted((json.load(open(p)) for p in glob.glob(out + ".*")), key=lambda x: x["rank"])
    assert [x["rank"] for x in recs] == [0, 1] and all(x["world"] == 2 for x in recs)
    # every rank sees the same (max) time, and it is the slow rank's
    assert abs(recs[0]["max"] - recs[1]["max"]) < 1e-9 and recs[0]["max"] >= recs[1]["mine"] - 1e-9
    assert abs(recs[0]["max"] - max(x["mine"] for x in recs)) < 1e-9
    # whole-job value = frames of all ranks / max time
    assert abs(recs[0]["value"] - 2 * 7 / recs[0]["max"]) < 1e-6
    # channel c lives on rank c % world (accelerator.cpp placement), every channel exactly once
    assert recs[0]["channels"] == [0, 2, 4, 6, 8] and recs[1]["channels"] == [1, 3, 5, 7, 9]
    # every rank drove every channel of its shard through the mixer on every step
    assert all(x["rendered"] == 7 * 5 and x["mode"] == ("gpu" if gpu else "dry") for x in recs)


def test_shard_helpers_single_process():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.shard_channels(5, 1) == [[0, 1, 2, 3, 4]]
    assert bench.shard_channels(3, 4) == [[0], [1], [2], []]
    assert bench.whole_job_value(8, 50, 0.5) == 800.0
