#!/usr/bin/env python3
"""HBM-side bytes per step from the two PMC passes of tools/profile_config.sh.
usage: tools/traffic_from_pmc.py <gpurun_out/prof_TAG dir> <TAG>  -> <dir>/<TAG>_traffic.json (+ stdout)
FETCH_SIZE (own pass) doubled, WRITE_SIZE exact (MI355X_MICROARCH.md, HBM section), summed over every
kernel of a step (pre-pass, hub pass, fixups included); torch's own kernels and the one-off scale
binding are left out."""
import collections
import csv
import json
import sys

out, tag = sys.argv[1], sys.argv[2]
STEPS = 4  # profile_config.sh runs the PMC passes with --steps 3 --warmup 1


def totals(path):
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not k.startswith("void hg::") and not k.startswith("hg::"):
            continue
        if "bind_scales" in k:
            continue
        per_kernel[k.split("(")[0].replace("void ", "")[:70]][r["Counter_Name"]] += float(r["Counter_Value"])
    return per_kernel


def bench_line(path):
    for line in open(path):
        if line.startswith('{"metric"'):
            return json.loads(line)
    raise SystemExit("no bench line in " + path)


f = totals("%s/pmc_fetch/f_counter_collection.csv" % out)
w = totals("%s/pmc_write/w_counter_collection.csv" % out)
kern = {}
tot_f = tot_w = hit = miss = 0.0
for k in sorted(set(f) | set(w)):
    fk = f.get(k, {}).get("FETCH_SIZE", 0.0) / STEPS
    wk = w.get(k, {}).get("WRITE_SIZE", 0.0) / STEPS
    kern[k] = {"fetch_x2_MB_per_step": 2 * fk / 1024, "write_MB_per_step": wk / 1024}
    tot_f += fk
    tot_w += wk
    hit += w.get(k, {}).get("TCC_HIT_sum", 0.0)
    miss += w.get(k, {}).get("TCC_MISS_sum", 0.0)
bench = bench_line("%s/pmc_fetch.log" % out)
entry = {"kernels": kern, "FETCH_SIZE_KB_per_step": tot_f, "WRITE_SIZE_KB_per_step": tot_w,
         "l2_hit_rate": hit / max(1.0, hit + miss),
         "bytes_per_step": (2.0 * tot_f + tot_w) * 1024.0,
         "algorithmic_bytes_per_step": bench["roofline"]["algorithmic_bytes_per_step"],
         "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes, "
                   "summed over every kernel of a step; FETCH_SIZE doubled (gfx950 reports half the bytes of "
                   "16 B/lane reads, MI355X_MICROARCH.md HBM section)"}
name = bench["config"]["workload"]  # bench.py's workload name (carries the ", weighted ..." suffix itself)
json.dump({name: entry}, open("%s/%s_traffic.json" % (out, tag), "w"), indent=1)
print(tag, name, json.dumps({k: entry[k] for k in ("bytes_per_step", "algorithmic_bytes_per_step", "l2_hit_rate")}))
for k, v in kern.items():
    print("   %-70s fetch(x2) %8.1f MB  write %8.1f MB" % (k, v["fetch_x2_MB_per_step"], v["write_MB_per_step"]))
