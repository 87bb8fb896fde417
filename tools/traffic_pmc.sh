#!/bin/bash
# HBM-side traffic of the dominant kernel for one bench configuration: FETCH_SIZE in its own pass
# (combining it with other counters hung the profiler on this pool), WRITE_SIZE + TCC hit/miss in another.
# usage: tools/traffic_pmc.sh <tag> <bench.py flags...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic_$tag
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/traffic_$tag/f -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline "$@" > gpurun_out/traffic_$tag/f.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d gpurun_out/traffic_$tag/w -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline "$@" > gpurun_out/traffic_$tag/w.log 2>&1 || exit 1
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/traffic_$tag/?/x_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "copyBuffer" not in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print("$tag", k, "fetch(x2) %.1f MB  write %.1f MB  L2 hit %.3f" % (2 * m.get("FETCH_SIZE", 0) / 1024, m.get("WRITE_SIZE", 0) / 1024,
          m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0))))
PY
