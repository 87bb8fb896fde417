#!/bin/bash
# Quick A/B over the bench shapes: prints ms/step and roofline fraction per configuration.
# usage: [LIBS="a.so b.so"] tools/bench_set.sh [extra bench.py flags]
# With LIBS every configuration runs once per library, back to back on the same box
# (fresh boxes differ by a few percent, so only same-box numbers compare).
for a in "" "--weighted" "--shape citeseer" "--shape pubmed --replicas 64 --feat 128" "--feat 64" "--feat 128 --replicas 256"; do
  for lib in ${LIBS:-default}; do
    if [ "$lib" != default ]; then export HG_AGGR_LIB=$(realpath $lib); fi
    python bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline $a "$@" 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-44s %-18s %.4f ms  frac %.3f' % ('$a', '$(basename $lib)', d['ms_per_step'], d['roofline']['frac']))"
  done
done
