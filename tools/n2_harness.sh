#!/bin/bash
# The N > 1 harness of bench.py as the driver starts it (torch.distributed.run is only the launcher: the ranks import
# no torch), rehearsed with two ranks on ONE GPU (--share-gpu).  1: the headline (independent streams, TCP group).
# 2: a leg that needs RCCL - two ranks on one device cannot form a communicator, so the leg must fail in bounded time,
# the headline line must still be printed and, with --strict-legs, the exit status must be 3 (EXIT_LEG_FAILED), not a hang
# (without that flag the status is 0 and the line names the leg under "legs_failed").
OUT=${1:-gpurun_out/n2}
mkdir -p $OUT
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
timeout -k 10 240 $L --master-port 29611 bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 --no-one-stream --no-relaxed --no-c2-legs > $OUT/n2_headline.json 2> $OUT/n2_headline.err
echo "headline: exit $?" | tee $OUT/n2_status.txt
timeout -k 10 300 $L --master-port 29612 bench.py --gpus 2 --share-gpu --steps 1 --warmup 0 --no-relaxed --no-c2-legs --stream-points 200000 --stream-blobs 5000 --stream-timeout 30 --strict-legs > $OUT/n2_leg.json 2> $OUT/n2_leg.err
echo "leg on a shared GPU: launcher exit $? (1 expected: the launcher's code for failed ranks; the ranks themselves exit 3, see below)" | tee -a $OUT/n2_status.txt
python - $OUT <<'PY' | tee -a $OUT/n2_status.txt
import json, sys
o = sys.argv[1]
a = json.loads(open(o + "/n2_headline.json").read().strip().splitlines()[-1])
print("headline: n_gpus %d, %.1f M points/s, line of %d bytes, %s" % (a["n_gpus"], a["value"] / 1e6, len(json.dumps(a)), a["config"].get("streams")))
b = json.loads(open(o + "/n2_leg.json").read().strip().splitlines()[-1])
print("leg line: value %.1f M points/s, legs_failed %s, leg_errors %s" % (b["value"] / 1e6, json.dumps(b.get("legs_failed")), json.dumps(b.get("leg_errors"))[:300]))
PY
grep "exitcode" $OUT/n2_leg.err | tee -a $OUT/n2_status.txt
