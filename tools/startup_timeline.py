"""Per-window kernel timeline of the LAST repetition of tools/startup.py from a rocprofv3 kernel trace (repetitions are
separated by a 20 ms sleep).  Windows are delimited by k_commit_b; one line per window: end time, wall time since the previous
window's end, kernel time inside it, then per kernel `us/launches`.  Usage: startup_timeline.py <trace dir> [first] [last] [w,w,.. windows listed kernel by kernel: duration(+gap before it)]"""
import glob
import sys

import pandas as pd

d = sys.argv[1]
first, last = int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 1000
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
df["name"] = df["Kernel_Name"].str.replace(r"^void ", "", regex=True).str.replace(r"\(.*", "", regex=True).str.slice(0, 28)
starts, ends = df["Start_Timestamp"].values, df["End_Timestamp"].values
cut = 0
for i in range(1, len(df)):
    if starts[i] - ends[:i].max() > 10_000_000:
        cut = i
st = df.iloc[cut:]
st = st[~st["name"].str.startswith("__amd")]
t0 = st["Start_Timestamp"].min()
prev_end, w, acc = t0, 0, {}
last_end = t0
detail = set(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else set()
seq = []
short = {"k_scan<20, true, true, true,": "dirty", "k_scan_u<20, 4>": "scan_u", "k_scan_p<20, 4, false>": "scan_p"}
for _, r in st.iterrows():
    k = short.get(r["name"], r["name"].replace("k_", ""))
    a = acc.setdefault(k, [0.0, 0])
    a[0] += (r["End_Timestamp"] - r["Start_Timestamp"]) / 1e3
    a[1] += 1
    seq.append("%s %.0f(+%.0f)" % (k, (r["End_Timestamp"] - r["Start_Timestamp"]) / 1e3, (r["Start_Timestamp"] - last_end) / 1e3 if seq or w else 0.0))
    last_end = r["End_Timestamp"]
    if r["name"].startswith("k_commit_b"):
        if first <= w < last:
            parts = ", ".join("%s %.0f/%d" % (n, v[0], v[1]) for n, v in sorted(acc.items(), key=lambda x: -x[1][0]))
            print("w%-3d end %7.3f ms  took %6.1f us  kernels %6.1f us | %s" % (
                w, (r["End_Timestamp"] - t0) / 1e6, (r["End_Timestamp"] - prev_end) / 1e3, sum(v[0] for v in acc.values()), parts))
        if w in detail:
            print("      " + " | ".join(seq))
        prev_end, w, acc = r["End_Timestamp"], w + 1, {}
        seq = []
