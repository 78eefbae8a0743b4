"""Per-kernel call counts and average / median durations of a rocprofv3 kernel trace.  Usage: kernel_avgs.py <dir> [name filter]"""
import glob
import sys

import pandas as pd

d = sys.argv[1]
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
df = pd.read_csv(f)
df["dur"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
df["name"] = df["Kernel_Name"].str.replace(r"^void ", "", regex=True).str.replace(r"\(.*", "", regex=True).str.slice(0, 48)
if len(sys.argv) > 2:
    df = df[df["name"].str.contains(sys.argv[2])]
agg = df.groupby("name")["dur"].agg(["count", "sum", "mean", "median", "max"]).sort_values("sum", ascending=False)
agg["sum"] /= 1e3
pd.set_option("display.width", 200)
print(agg.rename(columns={"sum": "total_ms", "mean": "avg_us", "median": "median_us", "max": "max_us"}).head(24).round(2).to_string())
