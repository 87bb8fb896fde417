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
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --steps 200 --warmup 20 $common "$@" > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
cp $out/stats/s_kernel_stats.csv $out/${tag}_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 $common "$@" > $out/pmc_fetch.log 2>&1 || { tail -5 $out/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $out/pmc_write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 $common "$@" > $out/pmc_write.log 2>&1 || { tail -5 $out/pmc_write.log; exit 1; }
python3 tools/traffic_from_pmc.py "$out" "$tag"
head -6 $out/${tag}_kernel_stats.csv
