S=$(date +%s); python bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; echo "rc $? elapsed $(( $(date +%s) - S )) s"
