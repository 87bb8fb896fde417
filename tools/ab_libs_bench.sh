#!/bin/bash
# Two builds of the library on a list of bench configurations, alternating, same box (HG_AGGR_LIB).
# usage (GPU box): tools/ab_libs_bench.sh <libA.so> <libB.so> [rounds]   (names under hypergef_amd/lib)
root=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $root
A=$1; B=$2; rounds=${3:-3}
run() { HG_AGGR_LIB=$root/hypergef_amd/lib/$1 timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --no-parity --no-configs --steps 100 --warmup 10 --detail /tmp/d.json "${@:2}" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   %-22s ms/step %.4f  frac %.4f' % ('$1', d['roofline']['avg_step_us'] / 1e3, d['roofline']['frac']))"; }
while read -r cfg; do
  [ -z "$cfg" ] && continue
  echo "== $cfg"
  for r in $(seq $rounds); do run $A $cfg; run $B $cfg; done
done <<CFG
${CONFIGS:---shape cora --replicas 1024 --feat 32 --weighted
--shape pubmed --replicas 256 --feat 32 --weighted
--shape cora --replicas 256 --feat 128 --weighted
--shape cora --replicas 1024 --feat 32
--shape pubmed --replicas 64 --feat 128}
CFG
