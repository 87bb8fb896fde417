#!/bin/bash
# Single-hypergraph latency (bench.py single_graph) under plan options.  usage: tools/single_sweep.sh <shape> <F> "<flags>" ...
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root; mkdir -p gpurun_out/sweep
shape=$1; F=$2; shift; shift
echo "== single $shape F=$F"
for v in "$@"; do
  timeout -k 10 200 python3 bench.py --shape $shape --replicas 2 --feat $F --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-parity $v > gpurun_out/sweep/run.log 2>&1 || { echo "FAILED: $v"; tail -3 gpurun_out/sweep/run.log; continue; }
  python3 - "$v" gpurun_out/sweep/run.log <<'PY'
import json, sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        d = json.loads(line)
        sg = d["single_graph"]
        print("   %-34s fused %.2f us  pull %.2f  push %.2f" % (sys.argv[1], sg["fused_us"], sg["pull_us"], sg["push_atomic_us"]))
PY
done
