#!/bin/bash
# Same-box A/B of the end-to-end driver between this tree and another checkout of the repo (e.g. a git worktree
# of an earlier commit, built in place).  usage (GPU box): tools/e2e_ab.sh <other tree> "<hgsys.py flags>" ...
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
other=$1; shift
mkdir -p $root/gpurun_out/e2e_ab
for fl in "$@"; do
  echo "== $fl"
  for rep in 1 2; do
    for t in $other $root; do
      out=$root/gpurun_out/e2e_ab/o.csv; rm -f $out
      (cd $t && timeout -k 10 200 python tools/hgsys.py $fl --output $out > /dev/null 2>&1) || echo "failed in $t"
      echo "   $(basename $t): $(cut -d, -f1,2,9,10 $out)"
    done
  done
done
