"""Per-kernel averages of the counters of one rocprofv3 --pmc pass (all dispatches with the largest grid of each kernel).
Usage: pmc_kernels.py <dir>"""
import glob
import sys

import pandas as pd

f = (glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0]
df = pd.read_csv(f)
df["name"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 34)
g = df.groupby(["name", "Dispatch_Id", "Counter_Name"])["Counter_Value"].sum().unstack()
t = df.groupby(["name", "Dispatch_Id"]).agg(s=("Start_Timestamp", "first"), e=("End_Timestamp", "first"), grid=("Grid_Size", "first"))
g["us"] = (t["e"] - t["s"]) / 1e3
g["grid"] = t["grid"]
rows = []
for name, sub in g.groupby(level=0):
    full = sub[sub["grid"] == sub["grid"].max()]
    m = full.mean()
    m["n"] = len(full)
    m.name = name
    rows.append(m)
out = pd.DataFrame(rows)
pd.set_option("display.width", 250)
pd.set_option("display.max_columns", 30)
print(out.round(1).to_string())
