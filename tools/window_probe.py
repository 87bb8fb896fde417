#!/usr/bin/env python3
"""Does the panels' re-read of the rows the materialisation pre-pass fetched hit the 256 MB Infinity Cache when the batch
is processed in windows?  A block-diagonal batch of K identical hypergraphs is aggregated as W calls over K / W
hypergraphs each (one plan for the window shape, X / Y row slices): every call runs its own pre-pass and then its
panels, so for a small enough window the panels find the pre-pass's rows in the cache (MI355X_MICROARCH.md, Infinity
Cache: a line stays resident while everything moved between its two uses fits in ~256 MiB -- about 2.7 x the window's
X slice here).  W = 1 is the whole batch in one call (what the library does).  The W calls are recorded into ONE
hipGraph (no host launch cost in the figure) on S streams dealt round-robin (S = 1: back to back; S = 2, 3: consecutive
windows overlap, which hides their launch gaps and tails).  Prints ms per whole batch.
usage: tools/window_probe.py [shape] [K] [F]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hypergef_amd import plan as planmod, synth
dev = "cuda:0"
shape = sys.argv[1] if len(sys.argv) > 1 else "pubmed"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
F = int(sys.argv[3]) if len(sys.argv) > 3 else 128
base = getattr(synth, shape + "_shape")()
X = torch.rand(base.N * K, F, device=dev)
Y = torch.empty_like(X)
balg = 4 * (2 * base.N * K * F + 2 * base.nnz * K + (base.M + base.N + 2) * K)
for W in [1, 2, 4, 8, 16, 32]:
    if K % W:
        continue
    k = K // W
    inc = synth.replicate_block_diagonal(base, k)
    ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
    plan = planmod.Plan.from_tensors(inc.N, ptr, ind)
    plan.prepare(F)
    n = inc.N
    for S in ([1] if W == 1 else [1, 2, 3]):
        wss = [torch.empty(max(plan.workspace_bytes(F), 256), dtype=torch.uint8, device=dev) for _ in range(S)]
        side = [torch.cuda.Stream() for _ in range(S)]

        def batch():
            cur = torch.cuda.current_stream()
            if S == 1:
                for w in range(W):
                    plan.aggregate(ptr, ind, X[w * n:(w + 1) * n], out=Y[w * n:(w + 1) * n], workspace=wss[0])
                return
            for st in side:
                st.wait_stream(cur)
            for w in range(W):
                with torch.cuda.stream(side[w % S]):
                    plan.aggregate(ptr, ind, X[w * n:(w + 1) * n], out=Y[w * n:(w + 1) * n], workspace=wss[w % S])
            for st in side:
                cur.wait_stream(st)
        for _ in range(2):
            batch()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            batch()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%s x%d F=%d: W=%2d windows of %3d hypergraphs (X slice %6.1f MB) on %d stream(s): %.4f ms per batch, frac %.3f of 8 TB/s"
              % (shape, K, F, W, k, n * F * 4 / 1e6, S, ms, balg / (ms * 1e-3) / 8e12), flush=True)
