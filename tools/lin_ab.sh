#!/bin/bash
# Same-box A/B of two library builds on the linear-epilogue shapes (tools/linear_probe.py rows).
# usage (GPU box): tools/lin_ab.sh <libA.so> <libB.so>
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
for rep in 1 2; do for lib in $1 $2; do
  echo "== $lib"
  HG_AGGR_LIB=$root/$lib timeout -k 10 500 python3 tools/linear_probe.py 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   %-14s %3d -> %3d  fused %.4f ms  two-step %.4f  aggr only %.4f  (%.1f TFLOP/s)' % (d['shape'], d['F_in'], d['F_out'], d['fused_ms'], d['two_step_ms'], d['aggr_only_Fin_ms'], d['mfma_tflops']))"
done; done
