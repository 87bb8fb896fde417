python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_full.json
python -c "import json; d=json.load(open('gpurun_out/bench_full.json')); print(d['ms_per_step'], d['roofline']['frac'], d['device_copy'], d['single_graph'])"
for tb in 16384 24576 32768 49152 65536; do for t in 4 8; do
python bench.py --shape pubmed --replicas 64 --feat 128 --steps 50 --warmup 5 --no-extras --no-cpu-baseline --variant fused --tile-bytes $tb --t-big $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('tile $tb tbig $t  %.4f ms frac %.3f' % (d['ms_per_step'], d['roofline']['frac']), d['fused_schedule'])"
done; done
