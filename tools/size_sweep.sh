#!/bin/bash
# Per-edge cost of the fused kernel against working-set size (cache-resident -> HBM).
for k in 16 32 64 128 256 512 1024 2048; do
  python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline --variant fused --replicas $k "$@" 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('replicas %5d  %.4f ms  %.3f ns/replica  frac %.3f' % ($k, d['ms_per_step'], d['ms_per_step']*1e6/$k, d['roofline']['frac']))"
done
