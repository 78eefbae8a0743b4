"""Per-window kernel timeline of one C2 step from a rocprofv3 kernel trace (run: rocprofv3 --kernel-trace --output-format csv
-d <dir> -- python bench.py --no-cpu-baseline --no-one-stream --no-relaxed --no-c2-legs --steps 1 --warmup 1).
Windows are delimited by k_commit_b; one line per window: when it ended (ms since the step's first kernel), how long it took
since the previous window's end, and the kernel time inside it by kernel (us, launches)."""
import glob
import sys

import pandas as pd

d = sys.argv[1]
first, last = int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 60
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
df["name"] = df["Kernel_Name"].str.replace(r"^void ", "", regex=True).str.replace(r"\(.*", "", regex=True).str.slice(0, 28)
# the last step: everything after the last-but-one offline phase (k_eps_neighbours)
eps = df.index[df["name"].str.startswith("k_eps_neighbours")].tolist()
start = eps[-2] + 1 if len(eps) >= 2 else 0
end = eps[-1] if eps else len(df)
st = df.iloc[start:end]
st = st[~st["name"].str.startswith("__amd")]
t0 = st["Start_Timestamp"].min()
prev_end, w, acc = t0, 0, {}
for _, r in st.iterrows():
    k = r["name"]
    a = acc.setdefault(k, [0.0, 0])
    a[0] += (r["End_Timestamp"] - r["Start_Timestamp"]) / 1e3
    a[1] += 1
    if k.startswith("k_commit_b"):
        if first <= w < last:
            parts = ", ".join("%s %.0f/%d" % (n.replace("k_", ""), v[0], v[1]) for n, v in sorted(acc.items(), key=lambda x: -x[1][0]))
            print("w%-3d end %7.3f ms  took %6.1f us  kernels %6.1f us | %s" % (
                w, (r["End_Timestamp"] - t0) / 1e6, (r["End_Timestamp"] - prev_end) / 1e3, sum(v[0] for v in acc.values()), parts))
        prev_end, w, acc = r["End_Timestamp"], w + 1, {}
