#!/usr/bin/env python3
"""bench.py -- composited frames/s of the HIP image mixer on BASELINE.json's headline configuration.

A step = one pass of the hot path: one channel composites one frame of the workload (default: configs[2], the
2160p50 16-layer ycbcr 4:2:2 16-bit stack with key layers, chroma key and levels, BGRA16 out) through the same
calls core::mixer makes (push / visit / pop / render, core/mixer/mixer.cpp:62-92). Source planes are uploaded once
and stay resident in HBM (a still frame is re-visited every tick without re-upload in the reference too); the
frame stays in HBM as well (no readback) so `value` is the compositor's own rate. The PCIe-inclusive rate is
reported separately as `end_to_end_fps` (pinned H2D of every plane + D2H of the frame each step).

Multi-GPU: channels shard by GPU with no data-path collective: rank r drives its own channel on GPU
LOCAL_RANK; value = N frames per step-time (weak scaling). torch.distributed is only the barrier + max-reduce.

    python bench.py --gpus 1 --steps 200 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured float4 copy)

WORKLOADS = {
    "config3_16bit": lambda S: S.config3(bits=16),
    "config3_8bit": lambda S: S.config3(bits=8),
    "config2": lambda S: S.config2(),
    "config1": lambda S: S.config1(),
    "config4": lambda S: S.config4(),
}


def dist_setup(n_gpus):
    """Returns (rank, world, local_rank, barrier, allreduce_max). Single process when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and not os.environ.get("CGHIP_DIST_FORCE_GROUP"):  # (set: a one-rank group, to rehearse the RCCL path)
        return rank, world, local_rank, (lambda: None), (lambda x: x)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
    use_gpu = torch.cuda.is_available()
    ndev = torch.cuda.device_count() if use_gpu else 0
    if use_gpu:
        torch.cuda.set_device(local_rank % ndev)  # a rehearsal may run more ranks than the box has GPUs
    backend = os.environ.get("CGHIP_DIST_BACKEND", "nccl" if use_gpu else "gloo")
    try:  # the rank's device is named, not guessed from the rank number
        kw = {"device_id": torch.device("cuda", local_rank % ndev)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, **kw)
    except Exception:
        backend = "gloo"
        dist.init_process_group(backend=backend)
    dev = torch.device("cuda", local_rank % ndev) if (backend == "nccl") else torch.device("cpu")

    def barrier():
        if backend == "nccl":
            dist.barrier(device_ids=[local_rank % ndev])
        else:
            dist.barrier()

    def allreduce_max(x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    return rank, world, local_rank, barrier, allreduce_max


def shard_channels(n_channels, world):
    """Channel c is placed on rank / GPU c % world (accelerator::create_image_mixer's round-robin placement):
    channels are independent, so this is the whole of the multi-GPU decomposition."""
    return [[c for c in range(n_channels) if c % world == r] for r in range(world)]


def whole_job_value(world, steps, seconds):
    """Whole-job frames/s: every rank composites `steps` frames in the (max-over-ranks) timed region."""
    return world * steps / seconds


def device_sync():
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass


class Channel:
    """One video channel: a HIP image mixer plus its resident source frames."""

    def __init__(self, A, S, scene, gpu, channel_id):
        self.A, self.S, self.scene = A, S, scene
        self.mixer = A.Mixer(channel_id=channel_id, gpu=gpu)
        self.ops = []
        self.frames = []
        for op in scene["ops"]:
            if op[0] == "push":
                self.ops.append(("push", A.default_transform(**op[1])))
            elif op[0] == "visit":
                spec = op[1]
                fr = self.mixer.create_frame(spec["fmt"], S.materialize(spec), spec.get("coords"))
                self.frames.append(fr)
                self.ops.append(("visit", fr))
            else:
                self.ops.append(("pop", None))

    def submit(self, readback=False):
        m = self.mixer
        for kind, arg in self.ops:
            if kind == "push":
                m.push(arg)
            elif kind == "visit":
                m.visit_frame(arg)
            else:
                m.pop()
        sc = self.scene
        return m.render_async(sc["width"], sc["height"], sc["square_width"], sc["square_height"], sc["bits"],
                              sc["field_mode"], readback=readback)

    def close(self):
        for f in self.frames:
            self.mixer.destroy_frame(f)
        self.mixer.close()


def run_steps(ch, steps, in_flight=2, readback=False):
    """Issue `steps` frames keeping `in_flight` in the pipeline (the reference keeps 1-2, mixer.cpp:75-91).
    Returns the list of per-frame kernel durations (ms, hipEvents on the channel's compute stream). readback: every
    frame is copied to pinned host memory on the channel's readback stream (device::copy_async(texture),
    device.cpp:264-302); the host-side copy out of the pinned buffer is the consumer's business and not made."""
    pend, kernel_ms = [], []
    for _ in range(steps):
        pend.append(ch.submit(readback=readback))
        if len(pend) > in_flight:
            ch.mixer.get(pend.pop(0), want_output=False)
            kernel_ms.append(ch.mixer.last_kernel_ms)
    while pend:
        ch.mixer.get(pend.pop(0), want_output=False)
        kernel_ms.append(ch.mixer.last_kernel_ms)
    return kernel_ms


def run_steps_channels(chs, steps, in_flight=2):
    """run_steps for several channels of one GPU: every tick each channel submits one frame on its own stream
    (the reference ticks its channels on independent threads, video_channel.cpp:125-193); completions are collected
    per channel, oldest first. Returns all kernel durations (ms)."""
    if len(chs) == 1:
        return run_steps(chs[0], steps, in_flight)
    pend, kernel_ms = [[] for _ in chs], []
    for _ in range(steps):
        for c, ch in enumerate(chs):
            pend[c].append(ch.submit(readback=False))
        for c, ch in enumerate(chs):
            if len(pend[c]) > in_flight:
                ch.mixer.get(pend[c].pop(0), want_output=False)
                kernel_ms.append(ch.mixer.last_kernel_ms)
    for c, ch in enumerate(chs):
        while pend[c]:
            ch.mixer.get(pend[c].pop(0), want_output=False)
            kernel_ms.append(ch.mixer.last_kernel_ms)
    return kernel_ms


def end_to_end_fps(A, S, scene, gpu, steps):
    """PCIe-inclusive: every step re-creates every source frame (copy into pinned memory + H2D) and reads the frame
    back (D2H); two frames in flight like core::mixer (mixer.cpp:85-91)."""
    m = A.Mixer(channel_id=99, gpu=gpu)
    cache = {}
    W, H, bits = scene["width"], scene["height"], scene["bits"]
    prev = None
    for i in range(steps + 3):
        if i == 3:  # three pipelined frames first: the pinned and device pools grow to two frames in flight
            m.get(prev)
            prev = None
            t0 = time.perf_counter()
        S.feed_product(m, scene, cache)
        cur = m.render_async(W, H, bits=bits, field_mode=scene.get("field_mode", 0))
        if prev is not None:
            m.get(prev)
        prev = cur
    m.get(prev)
    dt = time.perf_counter() - t0
    m.close()
    return steps / dt


def cpu_baseline(S, scene, budget_s=12.0):
    """The oracle (scalar C restatement of the reference mixer, OpenMP over rows) on this box's host cores."""
    from oracle import oracle as O
    cores = min(16, os.cpu_count() or 1)  # the GPU box gives one GPU's share of 16 cores
    O.lib().cpuref_set_threads(cores)
    cache = {}
    for op in scene["ops"]:  # synthesise the inputs outside the timed region
        if op[0] == "visit":
            S._planes(op[1], cache)
    n, t0 = 0, time.perf_counter()
    while True:
        S.run_oracle(scene, "cpu", cache)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 3:
            break
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d full frame(s) of the same workload (%s), includes texture upload copies and readback copy"
                      % (n, scene["name"])}


SETTLE_S = 5.0


def native_topology(args):
    """BASELINE configs[3] / configs[4] in the product's own topology - ONE process, N GPUs, a native thread per
    channel, routed frames copied GPU to GPU over xGMI (tests/native/route_driver.cpp) - as a child process started
    BEFORE this one touches the GPU. One JSON line in bench.py's format; `value` is the aggregate frames/s."""
    import subprocess
    rank = int(os.environ.get("RANK", "0"))
    if rank != 0:  # under torch.distributed.run: the driver is one process for the whole node, rank 0 starts it
        return
    exe = os.path.join(ROOT, "tests", "native", "route_driver")
    if not os.path.exists(exe):
        raise SystemExit("tests/native/route_driver is missing: run __graft_entry__.build()")
    wl = "config5" if args.workload == "config5" else "config4"
    cmd = [exe, "--workload", wl, "--gpus", str(args.gpus), "--frames", str(args.steps)]
    if args.channels_per_gpu > 1 or wl == "config4":
        per_gpu = args.channels_per_gpu if args.channels_per_gpu > 1 else max(1, 32 // args.gpus)
        cmd += ["--channels", str(per_gpu * args.gpus)]
    if args.stand_in:
        cmd.append("--stand-in")
    if args.readback:
        cmd.append("--readback")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise SystemExit("route_driver failed (%d): %s" % (r.returncode, (r.stderr or r.stdout)[-800:]))
    d = json.loads(r.stdout.strip().splitlines()[-1])
    line = {
        "metric": "aggregate composited frames/s, %s (%d channels over %d GPU(s), one process)" % (wl, d["channels"], d["gpus"]),
        "value": d["frames_per_s"], "unit": "frames/s", "n_gpus": d["gpus"], "steps": args.steps, "warmup": 10 if wl == "config5" else 50,
        "ms_per_step": 1e3 / d["per_channel_fps"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": d["workload"], "channels": d["channels"], "topology": d["topology"],
                   "route_ring": d["route_ring"], "stand_in_devices": d["stand_in_devices"], "readback": d["readback"]},
        "peer_copy_GBps": d["peer_copy_GBps"], "peer_copies": d["peer_copies"], "peer_copy_bytes": d["peer_copy_bytes"],
        "peer_access": d["peer_access"], "per_gpu_fps": d["per_gpu_fps"], "per_channel_fps": d["per_channel_fps"],
        "roofline": None, "cpu_baseline": None,
    }
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="config3_16bit", choices=sorted(WORKLOADS) + ["config5", "config4_native"],
                    help="config5 / config4_native: one process, N GPUs, a native thread per channel, routes over xGMI "
                         "(tests/native/route_driver.cpp); the others: one channel (or --channels-per-gpu) per rank")
    ap.add_argument("--stand-in", action="store_true",
                    help="config5 / config4_native on a box with fewer GPUs than --gpus: the missing GPUs are private device "
                         "objects on GPU 0 (the whole routed path runs, over one GPU's memory)")
    ap.add_argument("--channels-per-gpu", type=int, default=1,
                    help="channels each rank drives on its GPU, one stream each (configs[3]: 32 channels over N GPUs)")
    ap.add_argument("--settle", type=float, default=SETTLE_S,
                    help="seconds of untimed load before the warmup steps (counter-collection passes use a short one)")
    ap.add_argument("--readback", action="store_true",
                    help="after the timed region, a second one of the same length with every frame read back to pinned "
                         "host memory on the channel's readback stream; reported as `readback`, never as `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    args = ap.parse_args()
    if args.workload in ("config5", "config4_native"):
        return native_topology(args)  # a child process, started before anything here initialises the GPU

    rank, world, local_rank, barrier, allreduce_max = dist_setup(args.gpus)
    import scenes as S
    import server_amd as A

    # torch (the barrier / synchronize plumbing) is imported and its HIP state initialised HERE, before any phase that is
    # measured or that conditions the GPU: on a fresh box the first `import torch` pages in for many seconds, and when
    # that happened inside the first device_sync(), between the warmup steps and the timed region, the GPU sat idle for
    # that long and the 20 timed launches that followed ran 0.21 -> 0.26 ms (the second process on the same box: 0.21).
    device_sync()
    # ... and so are the communicators: the first collective of a process group sets them up (seconds with RCCL), which
    # must not happen between the warmup steps and the timed region either
    barrier()
    allreduce_max(0.0)
    ndev = A.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the mixer has no CPU fallback)")
    gpu = local_rank % ndev
    scene = WORKLOADS[args.workload](S)
    ncha = max(1, args.channels_per_gpu)
    chs = [Channel(A, S, scene, gpu, channel_id=rank * ncha + c + 1) for c in range(ncha)]

    # The GPU needs seconds of load before its power management settles: the same launch takes 0.33 ms in the first 50
    # frames after idle (tests/gpu_occupancy.py), and the first bench process on a fresh box measured 0.284 / 0.290 ms
    # after 0.5 / 2 s of load where the next process measured 0.275 / 0.280; after 6 s both measure the same (0.279 /
    # 0.277). A channel composites for hours, so the sustained rate is the one to report. Untimed, before the W warmup
    # steps, reported as config.settle_s.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        run_steps_channels(chs, 20)
    run_steps_channels(chs, max(1, args.warmup))
    device_sync()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = run_steps_channels(chs, args.steps)
    device_sync()
    t1 = time.perf_counter()  # this rank's K steps are complete; the closing barrier's own latency (a collective of some
    barrier()                 # 0.1 ms against 20 steps of 0.2 ms) is not part of any rank's work
    dt = allreduce_max(t1 - t0)  # every rank started at the opening barrier: the job took as long as its slowest rank

    alg_bytes = S.algorithmic_bytes(scene)
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    srt = sorted(kernel_ms)
    tenth = max(1, len(kernel_ms) // 10)
    launch_stats = {"n": len(kernel_ms), "min": srt[0], "median": srt[len(srt) // 2], "p95": srt[min(len(srt) - 1, int(0.95 * len(srt)))],
                    "max": srt[-1], "first_decile_mean": sum(kernel_ms[:tenth]) / tenth,
                    "last_decile_mean": sum(kernel_ms[-tenth:]) / tenth}
    achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
    if ncha > 1:
        # launches of different channels overlap on the GPU, so a launch's own duration says little: price the GPU's
        # algorithmic bytes over the timed region instead
        achieved = alg_bytes * ncha * args.steps / dt / 1e9

    rb = None
    if args.readback and ncha == 1:
        # resident sources, every frame read back: the channel as the reference runs it (the adapter always reads
        # back). Kernel and D2H overlap on two streams, so the rate to expect is min(kernel rate, D2H rate).
        run_steps(chs[0], max(4, args.warmup), readback=True)
        device_sync()
        tr0 = time.perf_counter()
        rb_kernel_ms = run_steps(chs[0], args.steps, readback=True)
        device_sync()
        rdt = time.perf_counter() - tr0
        frame_bytes = scene["width"] * scene["height"] * 4 * (scene["bits"] // 8)
        rb = {"frames_per_s": args.steps / rdt, "ms_per_step": rdt / args.steps * 1e3,
              "avg_kernel_ms": sum(rb_kernel_ms) / len(rb_kernel_ms), "frame_bytes": frame_bytes,
              "d2h_GBps": frame_bytes * args.steps / rdt / 1e9,
              "note": "sources resident, frame read back to pinned memory every step on the readback stream; 2 in flight"}

    e2e = None
    if rank == 0 and world == 1 and not args.no_end_to_end:
        e2e = end_to_end_fps(A, S, scene, gpu, 8)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(S, scene)
    for ch in chs:
        ch.close()

    if rank == 0:
        # HBM traffic and VALU instruction count per launch are NOT measured in this run: they come from the committed
        # rocprofv3 --pmc passes of the same command (profiles/collect.sh); the line names the files
        def committed(name):
            path = os.path.join(ROOT, "profiles", name)
            try:
                d = json.load(open(path))
                return d.get(args.workload), d.get("sources", {}).get(args.workload, "profiles/" + name)
            except Exception:
                return None, None
        traffic, traffic_source = committed("pmc_traffic.json")
        n_valu, valu_source = committed("pmc_valu.json")
        valu = None
        if n_valu:
            # the vector-ALU side of the same launch: duration measured live;
            # peak = 1024 SIMDs x one wave64 VALU instruction per 2 cycles at 2.4 GHz
            peak = 1024 * 2.4e9 / 2 / 1e9
            ach = n_valu / (avg_kernel_ms * 1e-3) / 1e9
            valu = {"wave_instr_per_launch": n_valu, "source": valu_source, "achieved": ach, "peak": peak,
                    "unit": "G wave-instr/s", "frac": ach / peak, "note": "see DESIGN.md section 4"}
        line = {
            "metric": "composited frames/s per GPU (2160p, 16 layers) + % of HBM roofline",
            "value": whole_job_value(world * ncha, args.steps, dt),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%d, %d layers, %d-bit out" % (
                scene["name"], scene["width"], scene["height"],
                sum(1 for o in scene["ops"] if o[0] == "visit"), scene["bits"]),
                "channels_per_gpu": ncha, "frames_in_flight": 2, "settle_s": args.settle, "parallelism": "channel-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         # the kernel instance the planner's launch of this workload runs (cghip_kernels.hip: launcher;
                         # the same names are in profiles/r03_*_kernel_stats.csv)
                         "kernel": {"config4": "cghip_composite_fast_p<8>", "config2": "cghip_composite<8, false, true>",
                                    "config1": "cghip_composite_fast_m<8, 1>", "config3_8bit": "cghip_composite_fast<8>",
                                    "config3_16bit": "cghip_composite_fast_m<16, 8>"}.get(
                             args.workload, "cghip_composite_fast<%d>" % scene["bits"]),
                         "avg_kernel_ms": avg_kernel_ms, "launch_ms": launch_stats,
                         "algorithmic_bytes_per_launch": alg_bytes, "valu": valu},
            "cpu_baseline": cpu,
            "end_to_end_fps": e2e,
        }
        if rb is not None:
            line["readback"] = rb
        print(json.dumps(line), flush=True)

    if world > 1 or os.environ.get("CGHIP_DIST_FORCE_GROUP"):
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
