#!/usr/bin/env python3
"""Does keeping Xe in a small, reused (Infinity-Cache resident) buffer pay?
Runs the K-block batch as K/S slices of S blocks that share one plan (identical
structure) and ONE workspace, vs the unsliced run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from hypergef_amd import plan as planmod, synth
from perf_probe import timeit

dev = "cuda:0"
F = int(os.environ.get("FEAT", "32"))
K = int(os.environ.get("K", "1024"))
base = synth.cora_shape()
X = torch.rand(base.N * K, F, device=dev)
Y = torch.empty(base.N * K, F, device=dev)
for S in [K, 512, 256, 128, 64, 32]:
    if S > K:
        continue
    inc = synth.replicate_block_diagonal(base, S)
    ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
    plan = planmod.Plan.from_tensors(inc.N, ptr, ind)
    ws = torch.empty(max(plan.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)
    nsl = K // S
    Xs = [X[i * inc.N:(i + 1) * inc.N] for i in range(nsl)]
    Ys = [Y[i * inc.N:(i + 1) * inc.N] for i in range(nsl)]

    def step():
        for i in range(nsl):
            plan.aggregate(ptr, ind, Xs[i], out=Ys[i], workspace=ws)
    t = timeit(step, 40)
    balg = 4 * (2 * base.N * K * F + 2 * base.nnz * K + (base.M + 2 * base.N) * K)
    print(json.dumps({"K": K, "slice_blocks": S, "slices": nsl, "xe_MB": inc.M * F * 4 / 1e6, "us": t * 1e6,
                      "us_per_block": t * 1e6 / K, "frac_alg": balg / t / 8e12}), flush=True)
