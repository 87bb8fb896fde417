#!/bin/bash
# Panel-kernel ablations (tuning build, HG_FUSED_DEBUG bits: 1 no X loads, 2 no Y stores, 4 no hop 1, 8 no hop 2,
# 16 record copy only, 64 non-temporal stores) over feature widths, cora x1024 batch.  usage: tools/ablate_widths.sh "8 16 32"
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
export HG_AGGR_LIB=$root/hypergef_amd/lib/libhgaggr_tuning.so
for f in $1; do
  for d in 0 1 2 3 4 8 16 64; do
    ms=$(HG_FUSED_DEBUG=$d timeout -k 10 120 python3 bench.py --feat $f --steps 100 --warmup 10 --no-parity --no-extras --no-configs --no-cpu-baseline $2 2>/dev/null | python3 -c "import json,sys; print('%.4f' % json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][0])['ms_per_step'])")
    echo "F=$f debug=$d ms=$ms"
  done
done
