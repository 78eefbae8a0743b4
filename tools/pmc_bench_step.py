"""VALU instructions of ONE bench step counted on the bench itself: reduces a rocprofv3 --pmc pass over
`python3 bench.py --gpus 1 --steps S --warmup W --no-cpu-baseline --no-transfers --no-one-stream --no-relaxed --no-c2-legs`
(counters SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE) to per-kernel sums per step.  Every step of the bench is the
same deterministic work (reset, the same points, a window policy that is a function of device counters), so the counts
of the W + S steps of the pass are W + S times one step's.
Usage: pmc_bench_step.py <pmc dir> <bench detail json of that pass> <steps + warmup>
bench.py quotes the figures (roofline.achieved / frac / step_frac) only for the csrc/ digest and workload they were counted
on: the instruction-lanes are a property of the work, the time base is the quoting run's own HIP-event time."""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

d, detail_path, n_steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
df = pd.read_csv(f)
df["name"] = df["Kernel_Name"].map(bench.kernel_short)
g = df.groupby(["name", "Dispatch_Id", "Counter_Name"])["Counter_Value"].sum().unstack().reset_index()
t = df.groupby(["name", "Dispatch_Id"]).agg(s=("Start_Timestamp", "first"), e=("End_Timestamp", "first")).reset_index()
g = g.merge(t, on=["name", "Dispatch_Id"])
g["us"] = (g["e"] - g["s"]) / 1e3
detail = json.load(open(detail_path))
cfg = detail["config"]
kernels = {}
for name, sub in g.groupby("name"):
    insts = float(sub["SQ_INSTS_VALU"].sum())
    cycles = float(sub["GRBM_GUI_ACTIVE"].sum()) / 8.0  # the counter sums the 8 XCDs
    kernels[name] = {
        "launches_per_step": len(sub) / n_steps,
        "valu_instructions_per_step": insts / n_steps,                # wave-level instructions (SQ_INSTS_VALU)
        "instruction_lanes_per_step": insts * 64.0 / n_steps,        # x 64 lanes: what `peak` counts
        "us_per_step_under_pmc": float(sub["us"].sum()) / n_steps,
        "valu_busy_fraction": float(sub["SQ_ACTIVE_INST_VALU"].sum()) * 4.0 / 1024.0 / cycles if cycles > 0 else None,
    }


def is_scan(name):
    return name.startswith(("k_scan_u<", "k_scan_p<", "k_scan_p2<", "k_scan_p3<", "k_prefix16<", "k_seed16<", "k_scan_a<", "k_seed<", "k_seed_merge<", "k_missed")) or \
        (name.startswith("k_scan<") and ", false, " in name)  # (k_scan<DP, FILTER, POW2, DIRTY=false, NW>: the LDS-staged snapshot scan)


scan = {k: v for k, v in kernels.items() if is_scan(k)}
out = {
    "what": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE over bench.py itself (%d identical steps): VALU "
            "instructions of one step of the headline configuration, per kernel" % n_steps,
    "csrc_sha256": bench.csrc_digest(),
    "points": cfg["points"], "dim": cfg["dim"], "microclusters": cfg["microclusters"], "window": cfg["window"],
    "windows_per_step": cfg.get("windows_per_step"),
    "steps_counted": n_steps,
    "ms_per_step_under_pmc": detail["ms_per_step"],
    "snapshot_scan": {
        "kernels": sorted(scan),
        "launches_per_step": sum(v["launches_per_step"] for k, v in scan.items() if k.startswith(("k_scan_u<", "k_scan_p<", "k_scan_p2<", "k_scan_p3<", "k_scan<"))),
        "instruction_lanes_per_step": sum(v["instruction_lanes_per_step"] for v in scan.values()),
        "us_per_step_under_pmc": sum(v["us_per_step_under_pmc"] for v in scan.values()),
    },
    "all_kernels": {
        "launches_per_step": sum(v["launches_per_step"] for v in kernels.values()),
        "instruction_lanes_per_step": sum(v["instruction_lanes_per_step"] for v in kernels.values()),
        "us_per_step_under_pmc": sum(v["us_per_step_under_pmc"] for v in kernels.values()),
    },
    "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["instruction_lanes_per_step"])),
    "note": "SQ_INSTS_VALU counts wave-level VALU instructions; x 64 = instruction-lanes, the unit of the 39.3 T/s peak "
            "(256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz).  One-off kernels of the run (upload: finiteness check, transposed copy) "
            "are divided by the step count like the rest: they are ~0.1 % of a step.",
}
print(json.dumps(out, indent=1))
