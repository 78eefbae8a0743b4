"""The snapshot scan of a bench run as rocprofv3 saw it, beside the bench line's own HIP-event figure.

A timed scan launch of bench.py is either one k_scan_u / k_scan launch or, when the scan is pruned, the chain
k_seed -> k_seed_merge -> k_scan_p on one stream (seeded), k_scan_p -> k_missed -> the seeded chain over the missed points
(guessed thresholds), or k_scan_p alone (lean).  This sums the kernel-trace durations of all of them and divides by the
number of launches (window-level k_scan_p + k_scan_u + clean k_scan dispatches), which is what `roofline.avg_launch_us`
measures.
Usage: scan_chain_summary.py <rocprof dir> <bench json> [leg name]"""
import glob
import json
import sys

import pandas as pd

d = sys.argv[1]
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
df = pd.read_csv(f)
df["dur"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
name = df["Kernel_Name"]
parts = {"k_seed": name.str.contains("k_seed<"), "k_seed_merge": name.str.contains("k_seed_merge"),
         "k_scan_p": name.str.contains("k_scan_p<"), "k_scan_p2": name.str.contains("k_scan_p2<"), "k_scan_p3": name.str.contains("k_scan_p3", regex=False),
         "k_prefix16": name.str.contains("k_prefix16", regex=False), "k_missed": name.str.contains("k_missed"), "k_scan_u": name.str.contains("k_scan_u<"),
         "k_scan (LDS-staged, clean)": name.str.contains(r"k_scan<\d+, (?:true|false), (?:true|false), false", regex=True)}
out = {"kernels": {}}
total = 0.0
for k, m in parts.items():
    g = df[m]
    if len(g):
        out["kernels"][k] = {"calls": int(len(g)), "total_ms": float(g["dur"].sum() / 1e3), "avg_us": float(g["dur"].mean())}
        total += float(g["dur"].sum())
# launches as bench.py times them: a plain scan (which may carry a probe of the pruned chain: three small kernels more) or
# a pruned chain - seeded for the window, or a guessed-threshold scan + k_missed + the seeded chain for the missed points.
# Every seeded chain has one k_seed_merge: a pruned launch's (the window's, or the missed points' behind k_missed) or a
# probe's; a probe's k_seed splits the rows over 128 workgroups per point tile (grid y), a launch's over at most 16.
ks = df[parts["k_seed"]]
probes = int((ks["Grid_Size_Y"] > 64).sum()) if len(ks) else 0
out["probes"] = probes
# Every pruned launch has exactly one window-level k_scan_p; a k_missed is followed by one more (over the list of missed
# points, same launch), a probe has one of its own (inside a plain launch); lean guessed scans are a k_scan_p and nothing else.
calls = {k: out["kernels"].get(k, {}).get("calls", 0) for k in parts}
# (round 6: the window's pruned scan is k_scan_p3 - k_scan_p2 / k_scan_p behind knobs -, and the points a guessed threshold
# missed go through k_scan_u over their list: one small k_scan_u per k_missed; k_scan_p then only runs in probes)
missed_plain = calls["k_scan_p3"] + calls["k_scan_p2"] > 0
if missed_plain:
    chains = calls["k_scan_p3"] + calls["k_scan_p2"] + max(0, calls["k_scan_p"] - probes) + calls["k_scan_u"] - calls["k_missed"] + \
        calls["k_scan (LDS-staged, clean)"]
else:
    chains = calls["k_scan_p"] - calls["k_missed"] - probes + calls["k_scan_u"] + calls["k_scan (LDS-staged, clean)"]
out["scan_chains"] = chains
out["rocprof_avg_chain_us"] = total / chains if chains else None
line = json.load(open(sys.argv[2]))
if len(sys.argv) > 3:
    line = line.get(sys.argv[3], {})
rf = line.get("roofline") or {}
out["bench_line_launches"] = rf.get("launches")
out["bench_line_avg_launch_us"] = rf.get("avg_launch_us")
out["note"] = ("bench.py brackets every scan launch (the whole chain when pruned) with HIP events on its stream; the profiled run "
               "includes the warm-up step(s), the bench line's figures only the timed ones")
print(json.dumps(out, indent=1))
