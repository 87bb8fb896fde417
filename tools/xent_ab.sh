#!/bin/bash
# pull variant: hop 1's Xe rows with plain stores (library A) against streaming stores (library B, -DHG_XE_NT=1).
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
for rep in 1 2; do for lib in libhgaggr.so libhgaggr_xent.so; do
  echo "== $lib"; HG_AGGR_LIB=$root/hypergef_amd/lib/$lib timeout -k 10 400 python3 tools/pull_ab.py 2>&1 | grep -E "yelp|Mushroom|walmart|dblp" | cut -c1-110
done; done
for lib in libhgaggr.so libhgaggr_xent.so; do for wl in "--shape cora --replicas 1024 --feat 32 --variant pull" "--shape pubmed --replicas 64 --feat 128 --variant pull"; do
  HG_AGGR_LIB=$root/hypergef_amd/lib/$lib timeout -k 10 200 python3 bench.py $wl --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-extras --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   %-20s %-60s ms %.4f' % ('$lib', '$wl', d['ms_per_step']))"
done; done
