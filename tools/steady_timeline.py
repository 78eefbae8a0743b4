"""Timeline of the steady state: kernels of a few consecutive windows from a rocprofv3 kernel trace of tools/steady.py
(REPS=1): per kernel its queue, start relative to the first one shown, duration, grid.  Also the per-kernel totals of the
steady run alone (everything after the longest gap in the trace: the host prints between the runs).
Usage: steady_timeline.py <rocprof dir> [first kernel index within the steady run] [count]"""
import glob
import sys

import pandas as pd

d = sys.argv[1]
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
df["name"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.slice(0, 36)
gap = df["Start_Timestamp"].diff()
cut = int(gap.idxmax())
st = df.iloc[cut:].reset_index(drop=True)
st["dur"] = (st["End_Timestamp"] - st["Start_Timestamp"]) / 1e3
span = (st["End_Timestamp"].max() - st["Start_Timestamp"].min()) / 1e3
print("steady run: %d kernels, span %.1f us, busy %.1f us" % (len(st), span, st["dur"].sum()))
tot = st.groupby("name")["dur"].agg(["count", "sum", "mean"]).sort_values("sum", ascending=False)
print(tot.to_string())
a = int(sys.argv[2]) if len(sys.argv) > 2 else len(st) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
w = st.iloc[a:a + n]
t0 = w["Start_Timestamp"].min()
qs = {q: i for i, q in enumerate(sorted(w["Queue_Id"].unique()))}
for _, r in w.iterrows():
    print("q%d %9.1f +%7.1f  %-36s grid %s" % (qs[r["Queue_Id"]], (r["Start_Timestamp"] - t0) / 1e3, r["dur"], r["name"], r["Grid_Size"]))
