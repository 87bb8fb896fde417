#!/usr/bin/env python3
"""pull vs fused in the throughput regime: K-fold block-diagonal batches of the dataset shapes."""
import sys, torch
sys.path.insert(0, '.')
from hypergef_amd import synth, plan as planmod
dev = 'cuda:0'


def t_ms(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for name, K in (("NTU2012", 512), ("ModelNet40", 64), ("coauthor_cora", 512), ("house-committees", 256),
                ("Mushroom", 64), ("20newsW100", 64), ("walmart-trips", 8), ("coauthor_dblp", 32), ("zoo", 2048), ("yelp", 1), ("yelp", 4), ("pubmed", 64),
                ("citeseer", 1024)):
    inc = synth.replicate_block_diagonal(synth.allset_shape(name), K)
    ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
    for F in (32, 64):
        X = torch.rand(inc.N, F, device=dev)
        Y = torch.empty_like(X)
        pl = planmod.Plan.from_tensors(inc.N, ptr, ind)
        info = pl.prepare(F)  # before sizing the workspace: the fused schedule's partial rows count
        ws = torch.empty(max(pl.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)
        res = {v: t_ms(lambda: pl.aggregate(ptr, ind, X, out=Y, workspace=ws, variant=v)) for v in ("pull", "fused")}
        print("%-18s x%-5d F %3d nnz %8d auto=%-5s pull %.4f fused %.4f  entries/nnz %.2f n_mat/M %.2f hubs %d split/N %.3f" % (
            name, K, F, inc.nnz, pl.auto_variant(F), res["pull"], res["fused"],
            (info["member_entries"] + info["hub_entries"]) / inc.nnz, info["n_mat"] / inc.M, info["n_hub"],
            info["n_split"] / inc.N), flush=True)
