#!/bin/bash
# A/B on one box: headline + single-graph latency + weighted + pubmed x64 F=128, for a list of bench.py flag sets.
# usage (GPU box): tools/ab_headline.sh <outdir-under-gpurun_out> "<flags A>" "<flags B>" ...
out=$1; shift
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root; mkdir -p gpurun_out/$out
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== $i: $v"
  for cfg in "--shape cora --feat 32" "--shape cora --feat 32 --weighted" "--shape citeseer --feat 32" "--shape pubmed --replicas 64 --feat 128" "--shape cora --replicas 256 --feat 128" "--shape cora --feat 64"; do
    timeout -k 10 200 python3 bench.py $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-configs $v > gpurun_out/$out/run.log 2>&1 || { echo "FAILED: $cfg $v"; tail -3 gpurun_out/$out/run.log; continue; }
    python3 - "$cfg" gpurun_out/$out/run.log <<'PY'
import json, sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        d = json.loads(line)
        sg = d.get("single_graph", {})
        print("   %-44s ms %.4f frac %.3f  single fused %.2f us  parity %s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["frac"], sg.get("fused_us", 0), d.get("parity", {}).get("ok")))
PY
  done
done
