"""Kernel by kernel through a few windows of the last bench step of a rocprofv3 kernel trace: start (us since the first listed
kernel), duration, stream (queue id) - what runs beside what in the steady state.  Usage: window_detail.py <trace dir> <first window> <count>"""
import glob
import sys

import pandas as pd

d, w0, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
df["name"] = df["Kernel_Name"].str.replace(r"^void ", "", regex=True).str.replace(r"\(.*", "", regex=True).str.slice(0, 24)
eps = df.index[df["name"].str.startswith("k_eps_neighbours")].tolist()
start = eps[-2] + 1 if len(eps) >= 2 else 0
end = eps[-1] if eps else len(df)
st = df.iloc[start:end]
st = st[~st["name"].str.startswith("__amd")].reset_index(drop=True)
cb = st.index[st["name"].str.startswith("k_commit_b")].tolist()
a = cb[w0 - 1] + 1 if w0 > 0 else 0
b = cb[min(len(cb) - 1, w0 + n - 1)]
# (a lookahead scan of these windows may have started before the previous commit: include kernels that END after it)
t_a = st.loc[a, "Start_Timestamp"]
sel = st[(st["End_Timestamp"] >= t_a) & (st.index <= b)]
t0 = sel["Start_Timestamp"].min()
qs = {q: i for i, q in enumerate(sorted(sel["Queue_Id"].unique()))}
for _, r in sel.iterrows():
    print("%9.1f %7.1f  q%d %s%s" % ((r["Start_Timestamp"] - t0) / 1e3, (r["End_Timestamp"] - r["Start_Timestamp"]) / 1e3, qs[r["Queue_Id"]],
                                   "        " * qs[r["Queue_Id"]], r["name"]))
