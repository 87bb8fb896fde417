#!/bin/bash
# Round profile set for the default bench workload, written under gpurun_out/profiles_new/:
# the bench line, the rocprofv3 --kernel-trace --stats summary of the same command, and the two
# PMC passes (FETCH_SIZE; WRITE_SIZE + TCC hit/miss) the roofline's `traffic` figure comes from.
# usage (on the GPU box): tools/collect_profiles.sh <tag, e.g. r01>
tag=${1:-r01}
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
out=$root/gpurun_out/profiles_new; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
python3 bench.py > $out/${tag}_bench_default.log 2>&1 || exit 1
tail -1 $out/${tag}_bench_default.log > $out/${tag}_bench_default.json
rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $out/stats.log 2>&1 || exit 1
cp $out/stats/s_kernel_stats.csv $out/${tag}_bench_default_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $out/pmc_write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $out/pmc_write.log 2>&1 || exit 1
cp $out/pmc_fetch/f_counter_collection.csv $out/${tag}_bench_fused_pmc_fetch.csv
cp $out/pmc_write/w_counter_collection.csv $out/${tag}_bench_fused_pmc_write.csv
python3 - <<PY
import csv, json, collections
def avg(path, kernel):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
f = avg("$out/${tag}_bench_fused_pmc_fetch.csv", "fused_packed_kernel")
w = avg("$out/${tag}_bench_fused_pmc_write.csv", "fused_packed_kernel")
bench = json.load(open("$out/${tag}_bench_default.json"))
entry = {"kernel": "fused_packed_kernel", "FETCH_SIZE_KB": f["FETCH_SIZE"], "WRITE_SIZE_KB": w["WRITE_SIZE"],
         "TCC_HIT_sum": w["TCC_HIT_sum"], "TCC_MISS_sum": w["TCC_MISS_sum"],
         "l2_hit_rate": w["TCC_HIT_sum"] / (w["TCC_HIT_sum"] + w["TCC_MISS_sum"]),
         "bytes_per_launch": (2.0 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0,
         "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes; "
                   "FETCH_SIZE doubled (gfx950 reports half the bytes of 16 B/lane reads, MI355X_MICROARCH.md HBM section)"}
json.dump({bench["config"]["workload"]: entry}, open("$out/traffic.json", "w"), indent=1)
print(json.dumps(entry))
PY
head -4 $out/${tag}_bench_default_kernel_stats.csv
