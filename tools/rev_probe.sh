#!/bin/bash
# Diagnostic build: panels of each XCD's run in forward (bit 1024 = no-op) against reverse order (bit 2048) after the
# materialisation pre-pass -- do the rows the pre-pass read last come back from the L2 / Infinity Cache?
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
for wl in "$@"; do
  echo "== $wl"
  for rep in 1 2; do for dbg in 1024 2048; do
    HG_FUSED_DEBUG=$dbg HG_AGGR_LIB=$root/hypergef_amd/lib/libhgaggr_tuning.so timeout -k 10 200 python3 bench.py $wl --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-extras --no-parity 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   debug %-5s ms %.4f frac %.3f' % ('$dbg', d['ms_per_step'], d['roofline']['frac']))"
  done; done
done
