#!/bin/bash
# Diagnostic build: row loads in flight per lane in the panel kernel (HG_FUSED_U = 8 | 12 | 16), same box.
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
for wl in "$@"; do
  echo "== $wl"
  for rep in 1 2; do for u in 8 12; do
    HG_FUSED_U=$u HG_AGGR_LIB=$root/hypergef_amd/lib/libhgaggr_tuning.so timeout -k 10 200 python3 bench.py $wl --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-extras --no-parity 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   U %-3s ms %.4f frac %.3f' % ('$u', d['ms_per_step'], d['roofline']['frac']))"
  done; done
done
