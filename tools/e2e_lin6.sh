#!/bin/bash
# End to end at nhid = 128 on a batch of 256 cora-shape hypergraphs: the fused layers' matrix phase on fp32 MFMA and as bf16x6
# (tools/hgsys.py --linear-math), torch baseline beside them.  usage (GPU box): tools/e2e_lin6.sh [out.csv]
root=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; cd $root
out=${1:-gpurun_out/e2e_lin6.csv}; rm -f $out; log=${out%.csv}.log; : > $log
run() { echo "== $*" >> $log; python3 tools/hgsys.py "$@" --output $out >> $log 2>&1 || echo "FAILED: hgsys.py $*"; }
for m in HGNN UniGCNII; do
  for lm in f32 bf16x6; do run --model $m --backend hgsys --dname cora --replicas 256 --nhid 128 --nlayer 4 --nfeat 128 --epochs 20 --linear-math $lm; done
  run --model $m --backend torch --dname cora --replicas 256 --nhid 128 --nlayer 4 --nfeat 128 --epochs 20
done
cat $out; grep -i "acc\|loss" $log | tail -12
