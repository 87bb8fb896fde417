#!/bin/bash
# Memory-system counter passes for one bench configuration (separate rocprofv3 --pmc runs, as the
# MI355X guide prescribes).  At most four counters per pass, and FETCH_SIZE never beside other TCC counters (it takes
# three of the four TCC slots: round 2's gpurun_out/pmc_f8/p1 aborted with "Request exceeds the capabilities of the
# hardware to collect"); FETCH_SIZE / WRITE_SIZE passes live in tools/profile_config.sh.  usage: tools/pmc_passes.sh <outdir-under-gpurun_out> <bench.py flags...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pmc in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TA_BUSY_avr TCC_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace -d gpurun_out/$out/p$i -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline "$@" > gpurun_out/$out.p$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/$out/p*/x_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:34], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/$out/summary.txt", "w") as o:
    for k, v in agg.items():
        if "copyBuffer" in k[0]: continue
        o.write("%s grid=%s\n" % k)
        for c, x in sorted(v.items()):
            o.write("    %-36s %.4g\n" % (c, sum(x) / len(x)))
print(open("gpurun_out/$out/summary.txt").read())
PY
