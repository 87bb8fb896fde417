#!/bin/bash
# Round 4: LDS tile size of the fused panels at F = 128 (slots per panel = bytes / 512) on the three target shapes, one box.
# usage (GPU box): tools/tile_sweep_r04.sh > gpurun_out/tile_sweep.log
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
run() { timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --no-parity --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   ms/step %.4f  frac %.4f' % (d['roofline']['avg_step_us'] / 1e3, d['roofline']['frac']))"; }
for rep in 1 2; do
for cfg in "cora 256 128" "citeseer 256 128" "pubmed 64 128" "cora 1024 32" "cora 1024 64"; do
  set -- $cfg
  for tb in 0 8192 12288 20480 24576 32768; do
    echo "== $1 x$2 F=$3 tile-bytes $tb"; run --shape $1 --replicas $2 --feat $3 --tile-bytes $tb || exit 1
  done
done
done
