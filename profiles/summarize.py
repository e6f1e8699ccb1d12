#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh (under gpurun_out/prof_<tag>/) into the small summaries
kept under profiles/: <tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json, <tag>_pmc_traffic.json,
<tag>_pmc_valu.json and pmc_traffic.json (the per-launch HBM traffic bench.py reports as roofline.traffic)."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit("missing %s under %s/%s" % (suffix, src, sub))
    return hits[0]


def counters(sub):
    per = collections.defaultdict(list)
    kernel = None
    for r in csv.DictReader(open(find(sub, "_counter_collection.csv"))):
        if "cghip_composite" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kernel = r["Kernel_Name"]
    return {k: sum(v) / len(v) for k, v in per.items()}, kernel


shutil.copy(find("stats", "_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_stats.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
bench = json.load(open(os.path.join(src, "bench_stats.json")))
# bench.py runs an untimed clock-settle phase and W warmup steps before its K timed steps; the --stats average covers
# all of those launches. The timed region is the LAST K launches of the trace: average them separately.
trace = [r for r in csv.DictReader(open(find("stats", "_kernel_trace.csv"))) if "cghip_composite" in r["Kernel_Name"]]
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in trace]
k = bench["steps"]
json.dump({"launches_in_trace": len(durs), "avg_ns_all": sum(durs) / len(durs), "timed_steps": k,
           "avg_ns_last_timed_steps": sum(durs[-k:]) / len(durs[-k:]), "min_ns": min(durs), "max_ns": max(durs),
           "bench_avg_kernel_ms": bench["roofline"]["avg_kernel_ms"]},
          open(os.path.join(dst, tag + "_kernel_timed_region.json"), "w"), indent=1)
workload = bench["config"]["workload"].split(":")[0]

f, kernel = counters("fetch")
w, _ = counters("write")
fetch_kb, write_kb = f["FETCH_SIZE"], w["WRITE_SIZE"]
traffic = {
    "workload": workload, "kernel": kernel,
    "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
    "fetch_bytes_raw": fetch_kb * 1024, "fetch_bytes_corrected_x2": fetch_kb * 2048, "write_bytes": write_kb * 1024,
    "traffic_bytes_per_launch": fetch_kb * 2048 + write_kb * 1024,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE); FETCH_SIZE is in KB and doubled per MI355X_MICROARCH.md "
            "(gfx950 tallies 128-B read requests at 64 B; calibrated for 16 B/lane streaming loads, this kernel mixes "
            "2/4/8 B per lane, so the x2 figure is an upper bound: raw <= true <= x2). WRITE_SIZE equals the frame size.",
}
json.dump(traffic, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
def merge(name, value, source):
    """profiles/<name>: workload -> latest per-launch figure (+ which file it came from), what bench.py looks up"""
    path = os.path.join(dst, name)
    try:
        d = json.load(open(path))
    except Exception:
        d = {}
    d[workload] = value
    d.setdefault("sources", {})[workload] = source
    d.pop("source", None)
    json.dump(d, open(path, "w"), indent=1)


merge("pmc_traffic.json", traffic["traffic_bytes_per_launch"], "profiles/%s_pmc_traffic.json" % tag)

v, _ = counters("valu")
v["kernel"] = kernel
v["note"] = ("per launch, summed over the chip; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles; "
             "GRBM_GUI_ACTIVE is summed over the 8 XCDs (divide by 8 for shader cycles per launch)")
json.dump(v, open(os.path.join(dst, tag + "_pmc_valu.json"), "w"), indent=1)
merge("pmc_valu.json", v.get("SQ_INSTS_VALU"), "profiles/%s_pmc_valu.json" % tag)
print(json.dumps({"traffic": traffic["traffic_bytes_per_launch"], "valu": v.get("SQ_INSTS_VALU")}))
