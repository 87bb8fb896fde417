#!/bin/bash
# The six north_star target cells (cora / citeseer / pubmed shape x F = 32 / 128) on one box: ms/step, frac, schedule shape.
# usage (GPU box): tools/matrix_probe.sh [extra bench flags]
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
tools/sweep.sh "--shape cora --replicas 1024 --feat 32" "$*"
tools/sweep.sh "--shape citeseer --replicas 1024 --feat 32" "$*"
tools/sweep.sh "--shape pubmed --replicas 256 --feat 32" "$*"
tools/sweep.sh "--shape cora --replicas 256 --feat 128" "$*"
tools/sweep.sh "--shape citeseer --replicas 256 --feat 128" "$*"
tools/sweep.sh "--shape pubmed --replicas 64 --feat 128" "$*"
