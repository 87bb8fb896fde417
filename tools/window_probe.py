#!/usr/bin/env python3
"""Does the panels' re-read of the rows the materialisation pre-pass fetched hit the 256 MB Infinity Cache when the batch
is processed in windows?  A block-diagonal batch of K identical hypergraphs is aggregated as W back-to-back calls over
K / W hypergraphs each (one plan for the window shape, X / Y row slices): every call runs its own pre-pass and then its
panels, so for a window whose X slice is below ~128 MB the panels find the pre-pass's rows in the cache.  W = 1 is the
whole batch in one call (what the library does).  Prints ms per whole batch; run under rocprofv3 --pmc FETCH_SIZE for
the HBM-side bytes (WINDOW_ONLY=W restricts the run to one window count).
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
only = os.environ.get("WINDOW_ONLY")
for W in ([int(only)] if only else [1, 2, 4, 8, 16, 32]):
    if K % W:
        continue
    k = K // W
    inc = synth.replicate_block_diagonal(base, k)
    ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
    plan = planmod.Plan.from_tensors(inc.N, ptr, ind)
    plan.prepare(F)
    ws = torch.empty(max(plan.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)
    n = inc.N

    def batch():
        for w in range(W):
            plan.aggregate(ptr, ind, X[w * n:(w + 1) * n], out=Y[w * n:(w + 1) * n], workspace=ws)
    for _ in range(3):
        batch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3 if only else 20
    e0.record()
    for _ in range(reps):
        batch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    x_mb = n * F * 4 / 1e6
    balg = 4 * (2 * base.N * K * F + 2 * base.nnz * K + (base.M + base.N + 2) * K)
    print("%s x%d F=%d: W=%2d windows of %3d hypergraphs (X slice %6.1f MB, nnz %d): %.4f ms per batch, frac %.3f of 8 TB/s"
          % (shape, K, F, W, k, x_mb, inc.nnz, ms, balg / (ms * 1e-3) / 8e12), flush=True)
