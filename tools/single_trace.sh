#!/bin/bash
# Kernel durations (rocprofv3 kernel trace) of single-hypergraph aggregations: what the kernel itself takes
# vs the launch-to-launch time bench.py's `single_graph` reports.  usage (GPU box): tools/single_trace.sh <shape> <F>
shape=${1:-cora}; F=${2:-32}
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd /tmp && export TMPDIR=/tmp && cd $root
out=gpurun_out/single_$shape$F; mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace -d $out/t -o s --output-format csv -- python3 bench.py --shape $shape --replicas 1 --feat $F --steps 200 --warmup 20 --no-cpu-baseline --no-configs > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 - $out/t/s_kernel_trace.csv $out/run.log <<'PY'
import csv, sys, json, collections
d = collections.defaultdict(list)
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    d[r["Kernel_Name"].split("(")[0][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    if "hg::" in k:
        v = sorted(v)
        print("   %-72s n %5d  median %.2f us  p10 %.2f  p90 %.2f" % (k, len(v), v[len(v) // 2] / 1e3, v[len(v) // 10] / 1e3, v[len(v) * 9 // 10] / 1e3))
# gaps between consecutive fused kernels (start - previous end) inside the graph replays
f = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "fused_packed" in r["Kernel_Name"])
gaps = sorted(b[0] - a[1] for a, b in zip(f, f[1:]) if 0 <= b[0] - a[1] < 20000)
if gaps:
    print("   gap between consecutive fused kernels: median %.2f us  p10 %.2f  p90 %.2f" % (gaps[len(gaps) // 2] / 1e3, gaps[len(gaps) // 10] / 1e3, gaps[len(gaps) * 9 // 10] / 1e3))
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        print("   bench single_graph:", json.loads(line).get("single_graph"))
PY
