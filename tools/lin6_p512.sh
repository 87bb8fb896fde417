#!/bin/bash
# bf16x6 epilogue on 512-thread panels (64 rows / 96 slots, HG_LIN_PANEL512=1, diagnostic build) against the shipped 256-thread
# panels (32 rows / 48 slots), same box, alternating.  The switch exists in commit 4dd5c20 only (make tuning there); the tree after it carries no
# 512-thread epilogue.  usage (GPU box): tools/lin6_p512.sh
root=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $root
export PROBE_CASES=${PROBE_CASES:-0,1,2,3} HG_AGGR_LIB=$root/hypergef_amd/lib/libhgaggr_tuning.so
for round in 1 2 3; do
  echo "== 256-thread panels"; timeout -k 10 200 python3 tools/bf16x6_probe.py 2 2>&1 | grep -v amdgpu.ids | cut -c1-260
  echo "== 512-thread panels"; HG_LIN_PANEL512=1 timeout -k 10 200 python3 tools/bf16x6_probe.py 2 2>&1 | grep -v amdgpu.ids | cut -c1-260
done
