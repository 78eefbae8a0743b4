"""The validation kernels of the start-up stretch (tools/startup.py) from three rocprofv3 --pmc passes:
Usage: pmc_validation_summary.py <dir a: GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU>
                                 <dir b: GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES>
                                 <dir c: GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT>
                                 <points of the stretch> <repetitions>
Per kernel over ALL its dispatches that did work (longer than 4 us): launches and time per repetition, VALU / SALU / LDS
instructions per POINT of the stretch, VALU-busy share of the kernel's own run time (SIMD-cycles with a VALU instruction
active / 1024 SIMDs / cycles), wave time issuing / waiting to issue / parked on s_waitcnt."""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

POINTS, REPS = int(sys.argv[4]), int(sys.argv[5])
KERNELS = ("k_scan<", "k_dseed", "k_decide", "k_chain(", "k_chain_long", "k_commit_a", "k_commit_b", "k_claims", "k_scan_u<", "k_scan_p<")


def load(d):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    df = pd.read_csv(f)
    g = df.groupby(["Kernel_Name", "Dispatch_Id", "Counter_Name"])["Counter_Value"].sum().unstack()
    t = df.groupby(["Kernel_Name", "Dispatch_Id"]).agg(s=("Start_Timestamp", "first"), e=("End_Timestamp", "first"))
    g["us"] = (t["e"] - t["s"]) / 1e3
    return g.reset_index()


a, b, c = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
out = {"stretch": "first %d points of the C2 stream (1 M x 20, 5 000 microclusters) from an empty table, tools/startup.py, %d repetitions" % (POINTS, REPS),
       "csrc_sha256": bench.csrc_digest(), "kernels": {}}
for pat in KERNELS:
    def pick(df):
        m = df[df["Kernel_Name"].str.contains(pat, regex=False)]
        if pat == "k_scan<":  # the dirty scans only (template argument DIRTY = true)
            m = m[m["Kernel_Name"].str.contains(r"k_scan<\d+, (?:true|false), (?:true|false), true", regex=True)]
        return m[m["us"] > 4.0]
    ka, kb, kc = pick(a), pick(b), pick(c)
    if not len(ka) or not len(kb):
        continue
    cyc = ka["GRBM_GUI_ACTIVE"].sum() / 8.0
    k = {"launches_per_run": len(ka) / REPS, "ms_per_run": float(ka["us"].sum() / 1e3 / REPS), "avg_us": float(ka["us"].mean()),
         "valu_instructions_per_point": float(ka["SQ_INSTS_VALU"].sum() / REPS / POINTS),
         "salu_instructions_per_point": float(ka["SQ_INSTS_SALU"].sum() / REPS / POINTS),
         "valu_busy_fraction": float(ka["SQ_ACTIVE_INST_VALU"].sum() * 4.0 / 1024.0 / cyc),
         "wave_cycles_per_point": float(ka["SQ_WAVE_CYCLES"].sum() * 4.0 / REPS / POINTS),
         "wave_time_issuing": float(kb["SQ_ACTIVE_INST_ANY"].sum() / kb["SQ_WAVE_CYCLES"].sum()),
         "wave_time_waiting_to_issue": float(kb["SQ_WAIT_INST_ANY"].sum() / kb["SQ_WAVE_CYCLES"].sum()),
         "wave_time_parked_on_waitcnt": float(kb["SQ_WAIT_ANY"].sum() / kb["SQ_WAVE_CYCLES"].sum()),
         # waves resident on average: wave-cycles / (cycles x 1024 SIMDs)
         "avg_waves_per_simd": float(ka["SQ_WAVE_CYCLES"].sum() * 4.0 / cyc / 1024.0)}
    if len(kc):
        ccyc = kc["GRBM_GUI_ACTIVE"].sum() / 8.0
        k["lds_instructions_per_point"] = float(kc["SQ_INSTS_LDS"].sum() / REPS / POINTS)
        k["lds_array_busy_fraction"] = float(kc["SQ_LDS_IDX_ACTIVE"].sum() / 256.0 / ccyc)
    out["kernels"][pat.rstrip("<(") + (" (DIRTY=true)" if pat == "k_scan<" else "")] = k
out["note"] = ("three PMC passes over the same stretch; SQ_* count quad-cycles (x 4 = cycles); a kernel whose waves are parked on "
               "s_waitcnt most of their time with a VALU-busy share of a few per cent is bound by the latency of its dependent "
               "memory accesses and by how few waves it has, not by instruction issue")
print(json.dumps(out, indent=1))
