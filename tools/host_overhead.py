#!/usr/bin/env python3
"""Host cost of one operator call (launch-bound regime): calls per second the Python layer can issue for one
cora-shape hypergraph, against a bare torch op of similar device cost, and where the time goes (cProfile)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hypergef_amd as hg
from hypergef_amd import synth
dev = "cuda:0"
inc = synth.cora_shape()
hyperg = hg.HyperGraph.from_incidence(inc, dev, data_name="cora")
x = torch.randn(inc.N, 32, device=dev)
W = torch.ones(inc.M, 1, device=dev)
lin = torch.nn.Linear(32, 32, bias=False).to(dev)


def rate(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


with torch.no_grad():
    print("HGNNAggr (weighted, plan cached)      %.1f us/call host" % rate(lambda: hg.HGNNAggr(hyperg, x, hyperg.degE, hyperg.degV, W)))
    print("UniGNNConv (unweighted)               %.1f us/call host" % rate(lambda: hg.UniGNNConv(hyperg, x)))
    print("HGNNAggrLinear (fused layer)          %.1f us/call host" % rate(lambda: hg.HGNNAggrLinear(hyperg, x, lin.weight, hyperg.degE, hyperg.degV, W)))
    print("torch index_add_ two-hop baseline     %.1f us/call host" % rate(lambda: torch.zeros(inc.N, 32, device=dev).index_add_(0, hyperg.H_T_colind.long(), x[hyperg.H_T_colind.long()])))
    print("torch mm 2708x32x32                   %.1f us/call host" % rate(lambda: x @ lin.weight.t()))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        hg.HGNNAggr(hyperg, x, hyperg.degE, hyperg.degV, W)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
