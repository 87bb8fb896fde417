#!/bin/bash
# Round-4 profile set, written under gpurun_out/r04/: the default bench line, and for every configuration of the
# bench line's `configs` the rocprofv3 kernel stats plus the two PMC passes (tools/profile_config.sh), merged into
# one traffic.json.  usage (GPU box): tools/collect_r04.sh
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root; out=gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python3 bench.py --detail $out/r04_bench_detail.json > $out/bench_default.log 2>&1 || { tail -5 $out/bench_default.log; exit 1; }
tail -1 $out/bench_default.log > $out/r04_bench_default.json
run() { tag=$1; shift; tools/profile_config.sh $tag "$@" > $out/prof_$tag.log 2>&1 || { echo "profile $tag failed"; tail -3 $out/prof_$tag.log; return; }
        cp gpurun_out/prof_$tag/${tag}_kernel_stats.csv $out/r04_${tag}_kernel_stats.csv
        cp gpurun_out/prof_$tag/pmc_fetch/f_counter_collection.csv $out/r04_${tag}_pmc_fetch.csv
        cp gpurun_out/prof_$tag/pmc_write/w_counter_collection.csv $out/r04_${tag}_pmc_write.csv
        cp gpurun_out/prof_$tag/${tag}_traffic.json $out/traffic_$tag.json; tail -4 $out/prof_$tag.log; }
run cora32 --shape cora --replicas 1024 --feat 32
run citeseer32 --shape citeseer --replicas 1024 --feat 32
run pubmed32 --shape pubmed --replicas 256 --feat 32
run cora128 --shape cora --replicas 256 --feat 128
run citeseer128 --shape citeseer --replicas 256 --feat 128
run pubmed128 --shape pubmed --replicas 64 --feat 128
run pubmed128lin --shape pubmed --replicas 64 --feat 128 --linear-out 128
run pubmed128lin6 --shape pubmed --replicas 64 --feat 128 --linear-out 128 --linear-math bf16x6
run powerlaw64 --shape powerlaw --feat 64
run weighted --shape cora --replicas 1024 --feat 32 --weighted
python3 - <<'PY'
import glob, json
merged = {}
for f in sorted(glob.glob("gpurun_out/r04/traffic_*.json")):
    merged.update(json.load(open(f)))
json.dump(merged, open("gpurun_out/r04/traffic.json", "w"), indent=1)
for k, v in merged.items():
    print("%-95s %.3f GB moved / %.3f GB algorithmic = %.2fx, L2 hit %.2f" % (k, v["bytes_per_step"] / 1e9, v["algorithmic_bytes_per_step"] / 1e9,
          v["bytes_per_step"] / v["algorithmic_bytes_per_step"], v["l2_hit_rate"]))
PY
