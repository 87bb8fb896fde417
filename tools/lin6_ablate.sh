#!/bin/bash
# bf16x6 epilogue, stamped diagnostic instance: what the matrix phase's time is sensitive to (timing only, results wrong):
# STAMP_DEBUG = 32 (stamps) + 1024 no B loads in the loop / 2048 no A reads / 4096 no MFMAs.  usage (GPU box): tools/lin6_ablate.sh
root=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $root
for d in 32 1056 2080 4128 3104 7200; do echo "== STAMP_DEBUG=$d"; STAMP_DEBUG=$d STAMP_MATH=bf16x6 timeout -k 10 200 python3 tools/lin_stamp_probe.py 2>&1 | grep -v amdgpu.ids | grep "per call\|hop 1\|hop 2\|matrix phase\|total ticks"; done
