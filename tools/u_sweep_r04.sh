#!/bin/bash
# Round 4: row gathers in flight per lane (U) in the plain fused panels without materialised slots: 8 (shipped), 10, 12
# (diagnostic build, HG_FUSED_U).  usage (GPU box): tools/u_sweep_r04.sh > gpurun_out/u_sweep.log
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
export HG_AGGR_LIB=$root/hypergef_amd/lib/libhgaggr_tuning.so
run() { timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --no-parity --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   ms/step %.4f  frac %.4f' % (d['roofline']['avg_step_us'] / 1e3, d['roofline']['frac']))"; }
for rep in 1 2; do
for cfg in "cora 1024 32" "cora 256 128" "citeseer 1024 32" "citeseer 256 128" "cora 1024 64" "cora 1024 32 --weighted"; do
  set -- $cfg
  for u in 8 10 12; do
    echo "== $1 x$2 F=$3 $4 U=$u"; HG_FUSED_U=$u run --shape $1 --replicas $2 --feat $3 $4 || exit 1
  done
done
done
