#!/usr/bin/env python3
"""Per-launch durations of the dominant sweep kernel in a rocprofv3 kernel trace of `bench.py --steps K --warmup W`,
and the mean over the K timed launches - the figure bench.py's HIP events report (the *_kernel_stats.csv beside it
averages the W warm-up launches in as well).
    python tools/trace_timed.py <x_kernel_trace.csv> <K> [kernel-name substring] [bench line of that run (json)] \
        > profiles/<tag>_timed_launches.txt"""
import csv
import sys

path, k = sys.argv[1], int(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else "k_sweep_erm<float, 0"
rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
name = rows[0]["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0] if rows else pat
print("# %s: %d launches in the trace; duration of each in ms, in launch order" % (name, len(d)))
print(" ".join("%.3f" % x for x in d))
print("# mean over all launches (= AverageNs of the kernel-stats csv): %.4f ms" % (sum(d) / len(d)))
print("# mean over the last %d launches (the timed region): %.4f ms; min %.4f, max %.4f" % (k, sum(d[-k:]) / k, min(d[-k:]), max(d[-k:])))
if len(sys.argv) > 4:
    import json
    j = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
    r = j["roofline"]
    kk = r["kernels"]["sweep_erm"]
    print("# bench.py's HIP events in the same (profiled) run: mean %.4f ms over %d launches, first %.4f, last %.4f; %.1f it/s" % (
        kk["avg_ms"], kk["launches"], r["kernel_ms_first"], r["kernel_ms_last"], j["value"]))
