"""HBM-side traffic of the snapshot scan from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of
`bench.py --steps 1 --warmup 0 --no-cpu-baseline`), per scan launch as bench.py counts them: a launch is one k_scan_u /
clean k_scan dispatch or, when the scan is pruned, the chain k_seed -> k_seed_merge -> k_scan_p.  Records the SHA-256 of the
kernel sources (bench.csrc_digest): bench.py only quotes the figure for the kernels it was measured on.
Usage: pmc_summary.py <fetch_dir> <write_dir> <bench line of the fetch run> <bench line of the write run> > json"""
import glob
import json
import os
import sys

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

# (k_scan_p3 / k_prefix16: rocprofv3 leaves these template instances mangled - matched without the "<")
CHAIN = ("k_seed<", "k_seed_merge", "k_scan_p<", "k_scan_p2<", "k_prefix16", "k_scan_p3", "k_missed", "k_scan_u<")
HEADS = ("k_scan_u<",)


def totals(d, counter):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    df = pd.read_csv(f)
    df = df[df["Counter_Name"] == counter]
    per = {}
    chains = 0
    for pat in CHAIN:
        g = df[df["Kernel_Name"].str.contains(pat, regex=False)].groupby("Dispatch_Id")["Counter_Value"].sum()
        per[pat.rstrip("<")] = {"dispatches": int(len(g)), "KB_total": float(g.sum())}
        if pat in HEADS:
            chains += int(len(g))
    return per, chains


if __name__ == "__main__":
    # launches as bench.py counts (and times) them: the figure of the line each profiled run printed - a plain launch may
    # carry a probe of the pruned chain, a pruned launch is a chain of three or five kernels
    # (the FULL records of the two runs - bench.py's detail files -, not the compact stdout lines)
    line1 = json.load(open(sys.argv[3]))
    line2 = json.load(open(sys.argv[4]))
    n1, n2 = int(line1["roofline"]["launches"]), int(line2["roofline"]["launches"])
    fetch, _ = totals(sys.argv[1], "FETCH_SIZE")
    write, _ = totals(sys.argv[2], "WRITE_SIZE")
    fetch_kb = sum(v["KB_total"] for v in fetch.values()) / max(n1, 1)
    write_kb = sum(v["KB_total"] for v in write.values()) / max(n2, 1)
    out = {"points": int(line1["config"]["points"]), "dim": int(line1["config"]["dim"]), "window": int(line1["config"]["window"]),
           "csrc_sha256": bench.csrc_digest(), "scan_sha256": bench.scan_digest(),
           "kernel": "snapshot scan: pruned chains (k_scan_p with guessed thresholds + k_missed + k_seed / k_seed_merge / k_scan_p for the missed points, or the seeded chain for the whole window) and k_scan_u launches", "launches": n1,
           "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
           "k_scan_clean_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
           "per_kernel_FETCH_SIZE": fetch, "per_kernel_WRITE_SIZE": write,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `bench.py --steps 1 --warmup 0`; "
                   "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half of "
                   "a wide coalesced read; narrower accesses are uncalibrated, so this is an upper estimate); includes "
                   "Infinity-Cache hits; summed over the kernels of a scan launch (pruned: k_seed + k_seed_merge + k_scan_p) "
                   "and averaged over all scan launches of the run (the start-up phase uses short windows and plain k_scan_u "
                   "launches). Beyond the algorithmic bytes: the per-sub-range seed winners and candidate partials that "
                   "the next kernel merges, and the table prefixes every point tile's workgroups re-read from L2."}
    print(json.dumps(out, indent=1))
