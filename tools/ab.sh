#!/bin/bash
# usage: tools/ab.sh "lib1.so lib2.so" <bench.py flags...> : the same bench line once per library
libs=$1; shift
for lib in $libs $libs; do
  HG_AGGR_LIB=$(realpath $lib) python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-12s %.4f ms  %.2f Gedges/s  frac %.3f  %s' % ('$(basename $lib)', d['ms_per_step'], d['value']/1e9, d['roofline']['frac'], d['config'].get('resolved_variant','')))"
done
