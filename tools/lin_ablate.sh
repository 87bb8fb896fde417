#!/bin/bash
# Diagnostic build: where the linear-epilogue launch spends its time on pubmed x64, 128 -> 128.  HG_FUSED_DEBUG bits
# (timing only, results are wrong): 256 = no matrix work (rows leave as they are), 512 = no B-fragment loads.
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
for dbg in 1024 512 256 768; do
  echo "== HG_FUSED_DEBUG=$dbg"
  HG_FUSED_DEBUG=$dbg HG_AGGR_LIB=$root/hypergef_amd/lib/libhgaggr_tuning.so PROBE_FROM=${PROBE_FROM:-5} PROBE_ONLY=${PROBE_ONLY:-6} timeout -k 10 300 python3 tools/linear_probe.py 2>&1 | tail -${PROBE_TAIL:-1}
done
echo "== production library"
PROBE_FROM=${PROBE_FROM:-5} PROBE_ONLY=${PROBE_ONLY:-6} timeout -k 10 300 python3 tools/linear_probe.py 2>&1 | tail -${PROBE_TAIL:-1}
