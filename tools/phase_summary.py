"""Kernel time by kernel inside a time slice of a rocprofv3 kernel trace: phase_summary.py <dir> <from_ms> <to_ms>
(times relative to the first k_decide of the LAST online run in the trace)."""
import glob
import sys

import pandas as pd

f = (glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[0]
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
df["name"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 34)
# the last run starts at the last k_check_finite / k_transpose_points (upload) or the last cc_reset: use the largest gap
starts = df.index[df["name"].str.startswith("k_decide")]
gaps = df["Start_Timestamp"].diff().fillna(0)
cut = gaps[gaps > 2e6].index.max() if (gaps > 2e6).any() else 0
run = df.loc[cut:]
t0 = run["Start_Timestamp"].iloc[0]
lo, hi = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
sl = run[(run["Start_Timestamp"] - t0 >= lo) & (run["Start_Timestamp"] - t0 < hi)].copy()
sl["dur"] = (sl["End_Timestamp"] - sl["Start_Timestamp"]) / 1e3
g = sl.groupby("name")["dur"].agg(["count", "sum", "mean"]).sort_values("sum", ascending=False)
print("slice %.1f-%.1f ms of the last run: %d kernels, %.2f ms of kernel time, span %.2f ms" % (
    lo / 1e6, hi / 1e6, len(sl), sl["dur"].sum() / 1e3, (sl["End_Timestamp"].max() - sl["Start_Timestamp"].min()) / 1e6))
print(g.round(1).to_string())
