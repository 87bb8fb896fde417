#!/bin/bash
# bf16x6 epilogue: a seventh workgroup per CU?  Panels of 24 rows / 40 slots (three operand planes of 24 rows = 18 KB inside a
# 20 KB tile: 22 KB of LDS per workgroup) on a build with the seven-wave register budget and eight gathers in flight
# (libhgaggr_t7.so: -DHG_LIN_WAVES_STAGED32=7 -DHG_LIN_U32=8), against the shipped 32 rows / 48 slots at six waves, same box.
# (libhgaggr_tuning.so: make tuning; libhgaggr_t7.so: the same with -DHG_LIN_WAVES_STAGED32=7 -DHG_LIN_U32=8, built by hand)
# usage (GPU box): tools/lin6_occ.sh
root=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $root
export PROBE_CASES=${PROBE_CASES:-0,2}
run() { echo "== $1 rows_cap ${2:-default} slots pct ${3:-150}"; HG_AGGR_LIB=$root/hypergef_amd/lib/$1 HG_LIN_ROWS_CAP=$2 HG_LIN_SLOTS_PCT=$3 HG_PRINT_OCC=${OCC:-0} timeout -k 10 200 python3 tools/bf16x6_probe.py 2 2>&1 | grep -v amdgpu.ids | cut -c1-260; }
for round in 1 2; do
  run libhgaggr_tuning.so 32 150
  run libhgaggr_t7.so 32 150
  run libhgaggr_t7.so 24 167
  run libhgaggr_tuning.so 24 167
  run libhgaggr_t7.so 24 200
  run libhgaggr_t7.so 16 200
  run libhgaggr_t7.so 16 250
done
