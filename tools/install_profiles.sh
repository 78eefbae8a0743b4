#!/bin/bash
# Copies the summaries of a tools/profile_round.sh run into profiles/ under the round's names (what profiles/README.md lists).
# Usage: bash tools/install_profiles.sh <outdir> <round, e.g. r05>   (build container, after gpurun merged the output back)
set -e
O=${1:?outdir}; R=${2:?round}
P=profiles
cpif() { if [ -f "$1" ]; then cp "$1" "$2"; echo "  $2"; fi; }
cpif $O/bench.json $P/${R}_bench.json
cpif $O/bench_detail.json $P/${R}_bench_detail.json
cpif $O/bench_under_rocprof.json $P/${R}_bench_under_rocprof.json
cpif $O/bench_under_rocprof_nolookahead.json $P/${R}_bench_under_rocprof_nolookahead.json
cpif $O/kernel_summary.txt $P/${R}_kernel_summary.txt
cpif $O/kernel_summary_nolookahead.txt $P/${R}_kernel_summary_nolookahead.txt
cpif $O/trace/r_kernel_stats.csv $P/${R}_kernel_stats.csv
cpif $O/trace_nola/r_kernel_stats.csv $P/${R}_kernel_stats_nolookahead.csv
cpif $O/scan_chain.json $P/${R}_scan_chain.json
for LEG in one_stream_exact events_sharded_relaxed; do
  cpif $O/bench_leg_$LEG.json $P/${R}_bench_leg_$LEG.json
  cpif $O/kernel_summary_leg_$LEG.txt $P/${R}_kernel_summary_leg_$LEG.txt
  cpif $O/trace_$LEG/r_kernel_stats.csv $P/${R}_kernel_stats_leg_$LEG.csv
  cpif $O/scan_chain_leg_$LEG.json $P/${R}_scan_chain_leg_$LEG.json
done
cpif $O/pmc_traffic.json $P/${R}_pmc_traffic.json
for f in $O/pmc_valu_d*.json; do [ -f "$f" ] && cpif $f $P/${R}_$(basename $f); done
true
