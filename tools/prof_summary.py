"""Summarises a rocprofv3 kernel trace: per kernel, launches that did real work vs early-exit launches."""
import glob
import sys

import pandas as pd

d = sys.argv[1]
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
df = pd.read_csv(f)
df["dur"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
df["name"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.slice(0, 40)
rows = []
for name, g in df.groupby("name"):
    thr = max(4.0, 0.25 * g["dur"].quantile(0.9))
    work = g[g["dur"] >= thr]
    idle = g[g["dur"] < thr]
    rows.append((name, len(g), g["dur"].sum() / 1e3, len(work), work["dur"].mean() if len(work) else 0.0,
                 len(idle), idle["dur"].mean() if len(idle) else 0.0))
out = pd.DataFrame(rows, columns=["kernel", "calls", "total_ms", "work_calls", "work_avg_us", "noop_calls", "noop_avg_us"])
print(out.sort_values("total_ms", ascending=False).head(12).to_string(index=False))
span = (df["End_Timestamp"].max() - df["Start_Timestamp"].min()) / 1e6
print("span %.1f ms, busy %.1f ms" % (span, df["dur"].sum() / 1e3))
