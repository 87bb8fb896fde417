#!/bin/bash
# Profile set for ONE bench configuration, written under gpurun_out/prof_<tag>/:
#   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command
#   <tag>_traffic.json       HBM-side bytes per step from the PMC counters: FETCH_SIZE in its own pass,
#                            WRITE_SIZE + TCC hit/miss in another (MI355X_MICROARCH.md, HBM section:
#                            separate --pmc passes, FETCH_SIZE doubled on gfx950), summed over every
#                            kernel of a step (pre-pass, hub pass, fixups included)
# usage (GPU box): tools/profile_config.sh <tag> <bench.py flags...>
tag=$1; shift
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
out=$root/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
common="--no-extras --no-cpu-baseline --no-parity"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --steps 50 --warmup 5 $common "$@" > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
cp $out/stats/s_kernel_stats.csv $out/${tag}_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 $common "$@" > $out/pmc_fetch.log 2>&1 || { tail -5 $out/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $out/pmc_write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 $common "$@" > $out/pmc_write.log 2>&1 || { tail -5 $out/pmc_write.log; exit 1; }
python3 - "$out" "$tag" <<'PY'
import csv, json, collections, sys
out, tag = sys.argv[1], sys.argv[2]
STEPS = 4  # --steps 3 --warmup 1
def totals(path):
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "copyBuffer" in k or "bind_scales" in k or "elementwise" in k or "fill" in k.lower():
            continue
        per_kernel[k.split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
    return per_kernel
f = totals("%s/pmc_fetch/f_counter_collection.csv" % out)
w = totals("%s/pmc_write/w_counter_collection.csv" % out)
kern = {}
tot_f = tot_w = hit = miss = 0.0
for k in sorted(set(f) | set(w)):
    fk = f.get(k, {}).get("FETCH_SIZE", 0.0) / STEPS
    wk = w.get(k, {}).get("WRITE_SIZE", 0.0) / STEPS
    kern[k] = {"fetch_x2_MB_per_step": 2 * fk / 1024, "write_MB_per_step": wk / 1024}
    tot_f += fk; tot_w += wk
    hit += w.get(k, {}).get("TCC_HIT_sum", 0.0); miss += w.get(k, {}).get("TCC_MISS_sum", 0.0)
bench = json.loads(open("%s/pmc_fetch.log" % out).read().strip().splitlines()[-1])
entry = {"kernels": kern, "FETCH_SIZE_KB_per_step": tot_f, "WRITE_SIZE_KB_per_step": tot_w,
         "l2_hit_rate": hit / max(1.0, hit + miss),
         "bytes_per_step": (2.0 * tot_f + tot_w) * 1024.0,
         "algorithmic_bytes_per_step": bench["roofline"]["algorithmic_bytes_per_step"],
         "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes, "
                   "summed over every kernel of a step; FETCH_SIZE doubled (gfx950 reports half the bytes of "
                   "16 B/lane reads, MI355X_MICROARCH.md HBM section)"}
json.dump({bench["config"]["workload"]: entry}, open("%s/%s_traffic.json" % (out, tag), "w"), indent=1)
print(tag, json.dumps({k: entry[k] for k in ("bytes_per_step", "algorithmic_bytes_per_step", "l2_hit_rate")}))
for k, v in kern.items():
    print("   %-60s fetch(x2) %8.1f MB  write %8.1f MB" % (k, v["fetch_x2_MB_per_step"], v["write_MB_per_step"]))
PY
head -6 $out/${tag}_kernel_stats.csv
