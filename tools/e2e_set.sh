#!/bin/bash
# The end-to-end driver lines kept in profiles/rNN_e2e_driver.csv (training / inference seconds per epoch), default mode:
# a model over ONE launch-bound hypergraph replays its training step and its forward as hipGraphs (both backends alike,
# tools/hgsys.py); batches of 256 hypergraphs run eager.  rNN_e2e_driver_eager.csv: the single-hypergraph rows with
# --no-graph (host-bound on both sides).
out=${1:-gpurun_out/e2e_driver.csv}; rm -f $out
log=${out%.csv}.log; : > $log
run() { python tools/hgsys.py "$@" >> $log 2>&1 || echo "FAILED: hgsys.py $*"; }
for m in HGNN UniGIN UniGCNII; do for b in hgsys torch; do
  run --model $m --backend $b --dname cora --epochs 100 --output $out
done; done
for b in hgsys torch; do run --model HGNN --backend $b --dname pubmed --nhid 128 --epochs 100 --output $out; done
for b in hgsys torch; do run --model HGNN --backend $b --dname cora --replicas 256 --epochs 30 --output $out; done
for b in hgsys torch; do run --model HGNN --backend $b --dname cora --replicas 256 --nhid 64 --nlayer 4 --nfeat 64 --epochs 30 --output $out; done
for m in UniGCNII UniGIN; do for b in hgsys torch; do
  run --model $m --backend $b --dname cora --replicas 256 --nhid 64 --nlayer 8 --nfeat 64 --epochs 20 --output $out
done; done
e=${out%.csv}_eager.csv; rm -f $e
for m in HGNN UniGIN UniGCNII; do for b in hgsys torch; do
  run --model $m --backend $b --dname cora --epochs 100 --no-graph --output $e
done; done
for b in hgsys torch; do run --model HGNN --backend $b --dname pubmed --nhid 128 --epochs 100 --no-graph --output $e; done
grep -i "graph capture\|FAILED" $log
cat $out; echo "-- single hypergraphs with --no-graph (eager)"; cat $e
