#!/bin/bash
# Power-law config under rocprofv3 --stats for a list of plan-option variants (one line per kernel).
# usage (GPU box): tools/pl_variants.sh <outdir-under-gpurun_out> "<flags of variant 1>" "<flags of variant 2>" ...
out=$1; shift
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p gpurun_out/$out
i=0
for v in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/$out/v$i -o s --output-format csv -- python3 bench.py --shape powerlaw --feat 64 --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-parity $v > gpurun_out/$out/v$i.log 2>&1 || { echo "variant $i ($v) failed"; tail -5 gpurun_out/$out/v$i.log; exit 1; }
  echo "== variant $i: $v"
  python3 - gpurun_out/$out/v$i.log gpurun_out/$out/v$i/s_kernel_stats.csv <<'PY'
import json, sys, csv
for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        d = json.loads(line)
        print("   ms/step %.3f  frac %.3f  sched %s" % (d["ms_per_step"], d["roofline"]["frac"], {k: d["fused_schedule"][k] for k in ("cap", "vdeg_max", "panels", "n_mat", "n_hub", "n_split", "member_entries", "hub_entries")} if d["fused_schedule"] else None))
for r in csv.DictReader(open(sys.argv[2])):
    if "copyBuffer" in r["Name"]: continue
    print("   %-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
