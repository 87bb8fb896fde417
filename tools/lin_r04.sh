#!/bin/bash
# Round 4, the linear epilogue (BASELINE config 3's MFMA path): the epilogue's own panel schedule (slots per panel as a
# percentage of its rows, rows per panel) and where the launch spends its time.  Diagnostic build; HG_FUSED_DEBUG bits
# (timing only, results are wrong): 1 = every X gather returns zeros without touching memory, 256 = no matrix work,
# 512 = no B-fragment loads.   usage (GPU box): tools/lin_r04.sh > gpurun_out/lin_r04.log
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
export HG_AGGR_LIB=$root/hypergef_amd/lib/libhgaggr_tuning.so
run() { timeout -k 10 300 python3 tools/linear_probe.py 2>&1 | tail -${PROBE_TAIL:-1}; }
for cfg in "5 6" "2 3" "1 2"; do
  set -- $cfg; export PROBE_FROM=$1 PROBE_ONLY=$2
  for pct in 0 125 150 200; do echo "== cfg $1 HG_LIN_SLOTS_PCT=$pct"; HG_LIN_SLOTS_PCT=$pct run || exit 1; done
  echo "== cfg $1 rows_cap 16, pct 150"; HG_LIN_SLOTS_PCT=150 HG_LIN_ROWS_CAP=16 run || exit 1
  echo "== cfg $1 rows_cap 16, pct 200"; HG_LIN_SLOTS_PCT=200 HG_LIN_ROWS_CAP=16 run || exit 1
done
export PROBE_FROM=5 PROBE_ONLY=6
for pct in 0 150; do for dbg in 1 256 257 512; do
  echo "== pubmed x64 128->128, HG_LIN_SLOTS_PCT=$pct HG_FUSED_DEBUG=$dbg"; HG_LIN_SLOTS_PCT=$pct HG_FUSED_DEBUG=$dbg run || exit 1
done; done
