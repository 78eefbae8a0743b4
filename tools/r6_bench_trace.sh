#!/bin/bash
# Round 6: kernel trace of ONE bench step (the headline configuration), per-kernel sums, the window timeline and the tail
# (offline phase + exports) kernel by kernel.
set -o pipefail
OUT=${1:-gpurun_out/r6bt}
mkdir -p $OUT
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o r -- python3 bench.py --gpus 1 --steps 2 --warmup 2 --no-cpu-baseline --no-transfers --no-one-stream --no-relaxed --no-c2-legs > $OUT/bench.txt 2>&1 || exit 1
python3 tools/timeline.py $OUT/trace 0 80 > $OUT/timeline.txt
python3 - $OUT/trace > $OUT/tail.txt <<'PY'
import glob, sys
import pandas as pd
d = sys.argv[1]
f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
df["name"] = df["Kernel_Name"].str.replace(r"^void ", "", regex=True).str.replace(r"\(.*", "", regex=True).str.slice(0, 40)
# the last step: from the last k_commit_b on = what follows the online phase; and the step's first kernels (reset, upload ..)
cb = df.index[df["name"].str.startswith("k_commit_b")].tolist()
eps = df.index[df["name"].str.startswith("k_eps_neighbours")].tolist()
last = cb[-1]
t0 = df.loc[last, "End_Timestamp"]
print("after the last commit of the last step (us since that commit's end: start, duration, gap before):")
prev = t0
for i in range(last + 1, len(df)):
    r = df.loc[i]
    print("%9.1f %8.1f %8.1f  %s" % ((r["Start_Timestamp"] - t0) / 1e3, (r["End_Timestamp"] - r["Start_Timestamp"]) / 1e3, (r["Start_Timestamp"] - prev) / 1e3, r["name"]))
    prev = r["End_Timestamp"]
# between the previous step's offline phase and this step's first commit
if len(eps) >= 2:
    a = eps[-2]
    first_cb = [i for i in cb if i > a][0]
    print("\nfrom the previous step's k_eps_neighbours to this step's first commit:")
    t0 = df.loc[a, "Start_Timestamp"]
    prev = t0
    for i in range(a, first_cb + 1):
        r = df.loc[i]
        print("%9.1f %8.1f %8.1f  %s" % ((r["Start_Timestamp"] - t0) / 1e3, (r["End_Timestamp"] - r["Start_Timestamp"]) / 1e3, (r["Start_Timestamp"] - prev) / 1e3, r["name"]))
        prev = r["End_Timestamp"]
PY
python3 tools/kernel_avgs.py $OUT/trace > $OUT/kernel_avgs.txt
python3 tools/window_detail.py $OUT/trace ${WDETAIL:-45} 2 > $OUT/window_detail.txt
find $OUT -name "*kernel_trace.csv" -delete
cut -c1-300 $OUT/bench.txt | tail -2; cat $OUT/tail.txt | head -120
