#!/bin/bash
# bf16x6 epilogue, matrix phase variants on one box: d1 = B fragments one step ahead, loaded after hop 2 (the first form);
# d3old = three steps of B in registers, the first three loaded before hop 2, A fragments read where used; libhgaggr.so = two
# steps of B + A fragments read in place three to five MFMAs ahead (shipped); d3pipe = the same with three steps (spills).
# The variant libraries are builds of intermediate trees (d1: the first form, commit b4c1d1d; d3old: -DHG_SPLIT_DEPTH=3
# -DHG_SPLIT_APIPE=0; d3pipe: -DHG_SPLIT_DEPTH=3), copied to hypergef_amd/lib/ under those names; LIBS="a.so b.so" runs any pair.
# usage (GPU box): tools/lin6_depth.sh
root=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $root
export PROBE_CASES=${PROBE_CASES:-0,1,2}
run() { echo "== $1"; HG_AGGR_LIB=$root/hypergef_amd/lib/$1 timeout -k 10 200 python3 tools/bf16x6_probe.py 2 2>&1 | grep -v amdgpu.ids | cut -c1-260; }
for round in 1 2 3; do for l in ${LIBS:-libhgaggr_d1.so libhgaggr_d3old.so libhgaggr.so libhgaggr_d3pipe.so}; do run $l; done; done
