#!/bin/bash
# Arbitrary PMC counters for one bench configuration, one rocprofv3 pass per counter group (counters only,
# --kernel-trace, never a sys/hip trace beside --pmc); prints the mean per launch for every hg:: kernel.
# usage (GPU box): tools/pmc_counters.sh <tag> "<bench flags>" "<counters of pass 1>" "<counters of pass 2>" ...
tag=$1; wl=$2; shift; shift
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
out=$root/gpurun_out/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 ${PMC_TIMEOUT:-90} rocprofv3 --pmc $grp --kernel-trace -d $out/p$i -o c --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-parity --no-configs $wl > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; continue; }
  python3 - $out/p$i/c_counter_collection.csv <<'PY'
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "hg::" not in k or "bind_scales" in k:
        continue
    a = acc[k.split("(")[0].replace("void ", "")[:60]][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print("   %-62s %s" % (k, "  ".join("%s=%.4g" % (c, v[0] / v[1]) for c, v in sorted(acc[k].items()))))
PY
done
