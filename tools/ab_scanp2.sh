cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in 1 0; do echo "== SCANP2=$v"; CHRONOCLUST_HIP_SCANP2=$v REPS=3 python3 tools/steady.py 2>&1 | grep "steady run [12]"; done
done
for v in 1 0 1 0; do echo "== bench SCANP2=$v"; CHRONOCLUST_HIP_SCANP2=$v python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-transfers --no-one-stream --no-relaxed --no-c2-legs 2>/dev/null | cut -c1-140; done
