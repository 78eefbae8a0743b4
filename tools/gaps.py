"""Where the wall time of a kernel trace goes: per kernel its busy time, and the idle time of the device between
consecutive kernels (end of one -> start of the next, over all streams: the union of busy intervals).
Usage: gaps.py <rocprofv3 --kernel-trace dir> [skip_first_ms]"""
import glob
import sys

import pandas as pd

d = sys.argv[1]
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
df["name"] = df["Kernel_Name"].str.replace(r"^void ", "", regex=True).str.replace(r"\(.*", "", regex=True).str.slice(0, 44)
# the last repetition only: everything after the last k_reset-like gap > 2 ms is one online run
starts = df["Start_Timestamp"].values
ends = df["End_Timestamp"].values
cut = 0
for i in range(1, len(df)):
    if starts[i] - ends[:i].max() > 2_000_000:
        cut = i
df = df.iloc[cut:].reset_index(drop=True)
df["dur"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
span = (df["End_Timestamp"].max() - df["Start_Timestamp"].min()) / 1e3
busy, cur_s, cur_e = 0.0, None, None
gaps = []
for s, e, nm in zip(df["Start_Timestamp"], df["End_Timestamp"], df["name"]):
    if cur_e is None:
        cur_s, cur_e = s, e
        continue
    if s > cur_e:
        busy += (cur_e - cur_s) / 1e3
        gaps.append(((s - cur_e) / 1e3, nm))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += (cur_e - cur_s) / 1e3
g = pd.DataFrame(gaps, columns=["gap_us", "before"])
print("last run: %d kernels, span %.2f ms, device busy (union) %.2f ms, idle between kernels %.2f ms in %d gaps (median %.1f us)" % (
    len(df), span / 1e3, busy / 1e3, g["gap_us"].sum() / 1e3, len(g), g["gap_us"].median()))
agg = df.groupby("name")["dur"].agg(["count", "sum", "mean", "median"]).sort_values("sum", ascending=False)
agg["sum"] /= 1e3
print(agg.rename(columns={"sum": "total_ms", "mean": "avg_us", "median": "median_us"}).head(16).round(2).to_string())
gg = g.groupby("before")["gap_us"].agg(["count", "sum", "mean"]).sort_values("sum", ascending=False)
gg["sum"] /= 1e3
print("idle time by the kernel that follows the gap:")
print(gg.rename(columns={"sum": "total_ms", "mean": "avg_us"}).head(12).round(2).to_string())
