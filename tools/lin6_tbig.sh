#!/bin/bash
# pubmed x64, 128 -> 128 with the bf16x6 matrix phase: the materialisation threshold (hyperedges with more members than t_big
# are summed once by the pre-pass, the rest recomputed in every panel that touches them).  usage (GPU box): tools/lin6_tbig.sh
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for round in 1 2; do for t in 0 4 6 12 16 24 32; do
  python3 bench.py --shape pubmed --replicas 64 --feat 128 --linear-out 128 --linear-math bf16x6 --t-big $t --steps 100 --warmup 10 \
    --no-extras --no-cpu-baseline --no-configs --no-parity --detail /tmp/d.json 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('t_big $t round $round: %.4f ms' % d['roofline']['avg_step_us'] if False else 't_big $t round $round: %.1f us' % d['roofline']['avg_step_us'])"
done; done
