#!/bin/bash
# The end-to-end driver lines kept in profiles/rNN_e2e_driver.csv (training / inference seconds per epoch).
out=${1:-gpurun_out/e2e_driver.csv}; rm -f $out
for m in HGNN UniGIN UniGCNII; do for b in hgsys torch; do
  python tools/hgsys.py --model $m --backend $b --dname cora --epochs 100 --output $out > /dev/null 2>&1
done; done
for b in hgsys torch; do python tools/hgsys.py --model HGNN --backend $b --dname pubmed --nhid 128 --epochs 100 --output $out > /dev/null 2>&1; done
for b in hgsys torch; do python tools/hgsys.py --model HGNN --backend $b --dname cora --replicas 256 --epochs 30 --output $out > /dev/null 2>&1; done
for b in hgsys torch; do python tools/hgsys.py --model HGNN --backend $b --dname cora --replicas 256 --nhid 64 --nlayer 4 --nfeat 64 --epochs 30 --output $out > /dev/null 2>&1; done
for m in UniGCNII UniGIN; do for b in hgsys torch; do
  python tools/hgsys.py --model $m --backend $b --dname cora --replicas 256 --nhid 64 --nlayer 8 --nfeat 64 --epochs 20 --output $out > /dev/null 2>&1
done; done
# launch-bound models (one dataset-sized hypergraph): inference as one hipGraph replay per forward
g=${out%.csv}_graph.csv; rm -f $g
for m in HGNN UniGIN UniGCNII; do for b in hgsys torch; do
  python tools/hgsys.py --model $m --backend $b --dname cora --epochs 100 --graph --output $g > /dev/null 2>&1
done; done
for b in hgsys torch; do python tools/hgsys.py --model HGNN --backend $b --dname pubmed --nhid 128 --epochs 100 --graph --output $g > /dev/null 2>&1; done
# the same models with the training step captured as well (forward, loss, backward, Adam: one replay per epoch)
t=${out%.csv}_graph_train.csv; rm -f $t
for m in HGNN UniGCNII; do for b in hgsys torch; do
  python tools/hgsys.py --model $m --backend $b --dname cora --epochs 100 --graph --graph-train --output $t 2>&1 | grep -i "graph capture" 
done; done
for b in hgsys torch; do python tools/hgsys.py --model HGNN --backend $b --dname pubmed --nhid 128 --epochs 100 --graph --graph-train --output $t 2>&1 | grep -i "graph capture"; done
cat $out; echo "-- with --graph (last column: hipGraph replay)"; cat $g
echo "-- with --graph --graph-train (both columns: hipGraph replay)"; cat $t
