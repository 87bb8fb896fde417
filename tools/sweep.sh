#!/bin/bash
# One bench configuration under a list of plan-option flag sets, same box: ms/step, frac, schedule shape.
# usage (GPU box): tools/sweep.sh "<workload flags>" "<flags 1>" "<flags 2>" ...
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$(cd $(dirname $0)/.. && pwd)
cd $root; mkdir -p gpurun_out/sweep
wl=$1; shift
echo "== workload: $wl"
for v in "$@"; do
  timeout -k 10 200 python3 bench.py $wl --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-extras $v > gpurun_out/sweep/run.log 2>&1 || { echo "FAILED: $v"; tail -3 gpurun_out/sweep/run.log; continue; }
  python3 - "$v" gpurun_out/sweep/run.log <<'PY'
import json, sys
for line in open(sys.argv[2]):
    if line.startswith('{"metric"'):
        d = json.loads(line)
        fs = d.get("fused_schedule") or {}
        print("   %-40s ms %.4f frac %.3f parity %s  %s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["frac"], d.get("parity", {}).get("ok"),
              {k: fs.get(k) for k in ("cap", "vdeg_max", "panels", "n_mat", "n_split", "member_entries")}))
PY
done
