#!/bin/bash
# Same-box A/B of two builds of libhgaggr.so (HG_AGGR_LIB): alternating runs of a list of workloads.
# usage (GPU box): tools/ab_lib.sh <libA.so> <libB.so> "<bench flags 1>" "<bench flags 2>" ...
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root; mkdir -p gpurun_out/ab
A=$1; B=$2; shift; shift
for wl in "$@"; do
  echo "== $wl"
  for rep in 1 2 3; do
    for lib in $A $B; do
      HG_AGGR_LIB=$root/$lib timeout -k 10 200 python3 bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-extras --no-parity > gpurun_out/ab/run.log 2>&1 || { echo "FAILED $lib"; tail -3 gpurun_out/ab/run.log; continue; }
      python3 - "$lib" gpurun_out/ab/run.log <<'PY'
import json, sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        d = json.loads(line)
        print("   %-40s ms %.4f frac %.3f" % (sys.argv[1].split('/')[-1], d["ms_per_step"], d["roofline"]["frac"]))
PY
    done
  done
done
