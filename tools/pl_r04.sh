#!/bin/bash
# Round 4, power-law config (|V| = 1M, |E| = 4M, F = 64): the hub pass beside the vertex panels on a second stream
# (HG_TWO_STREAM=1) and a 512-thread hub workgroup that leaves half the register file to panel workgroups
# (HG_HUB_THREADS=512).  Diagnostic build OF COMMIT 39c38fa: both switches lost and were removed from the tree afterwards
# (profiles/r04_experiments.md, section 4).  usage (GPU box, that tree): tools/pl_r04.sh > gpurun_out/pl_r04.log
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root
export HG_AGGR_LIB=$root/hypergef_amd/lib/libhgaggr_tuning.so
run() { timeout -k 10 400 python3 bench.py --shape powerlaw --feat 64 --no-extras --no-cpu-baseline --steps 50 --warmup 5 "$@" 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms/step %.4f  frac %.4f  parity %s' % (d['ms_per_step'], d['roofline']['frac'], d.get('parity')))"; }
for rep in 1 2; do
  echo "== baseline (one stream, 1024-thread hub workgroups)"; run --no-parity || exit 1
  echo "== HG_TWO_STREAM=1"; HG_TWO_STREAM=1 run --no-parity || exit 1
  echo "== HG_HUB_THREADS=512"; HG_HUB_THREADS=512 run --no-parity || exit 1
  echo "== HG_HUB_THREADS=512 HG_TWO_STREAM=1"; HG_HUB_THREADS=512 HG_TWO_STREAM=1 run --no-parity || exit 1
done
echo "== parity: HG_HUB_THREADS=512 HG_TWO_STREAM=1"; HG_HUB_THREADS=512 HG_TWO_STREAM=1 run || exit 1
echo "== parity: HG_TWO_STREAM=1"; HG_TWO_STREAM=1 run || exit 1
