#!/bin/bash
# One tuning-build environment knob over a list of values, per workload (same box).
# usage (GPU box): tools/knob_sweep.sh HG_FUSED_U "8 16 8 16" "<bench flags 1>" "<bench flags 2>" ...
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
export HG_AGGR_LIB=$root/hypergef_amd/lib/libhgaggr_tuning.so
knob=$1; vals=$2; shift; shift
for wl in "$@"; do
  echo "== $wl"
  for v in $vals; do
    ms=$(env $knob=$v timeout -k 10 150 python3 bench.py $wl --steps 100 --warmup 10 --no-parity --no-extras --no-configs --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print('%.4f' % json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][0])['ms_per_step'])")
    echo "   $knob=$v ms=$ms"
  done
done
