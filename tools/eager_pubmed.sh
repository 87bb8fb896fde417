#!/bin/bash
# Eager (no hipGraph) training / inference epoch of the 2-layer HGNN, nhid = 128, one pubmed-shape hypergraph: hgsys against
# the torch index_add_ baseline, three alternating runs (host-launch-bound: expect +-10 % between runs).
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
for rep in 1 2 3; do for b in hgsys torch; do
  python tools/hgsys.py --model HGNN --backend $b --dname pubmed --nhid 128 --epochs 100 2>/dev/null | grep "avg epoch time\|avg inference" | tr '\n' ' '; echo
done; done
