#!/bin/bash
# Same box: plain against streaming (nt) stores of Y over feature widths (two library builds).
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
for F in "$@"; do
  line="F=$F"
  for lib in libhgaggr_plainy.so libhgaggr.so libhgaggr_plainy.so libhgaggr.so; do
    r=$(HG_AGGR_LIB=$root/hypergef_amd/lib/$lib timeout -k 10 200 python3 bench.py --shape cora --replicas 1024 --feat $F --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-extras --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f/%.3f' % (d['ms_per_step'], d['roofline']['frac']))")
    line="$line  ${lib#libhgaggr}:$r"
  done
  echo "$line"
done
