#!/usr/bin/env python3
"""How fast can gather_rows_kernel gather 128-byte rows when the table is cache
resident?  M hyperedges of fixed size s over a small vertex set (table = N*F*4)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from hypergef_amd import plan as planmod, _lib
from perf_probe import timeit

dev = "cuda:0"
F = int(os.environ.get("FEAT", "32"))
for N, M, s in [(2708, 1_600_000, 3), (2708, 400_000, 12), (65536, 1_600_000, 3), (1 << 20, 1_600_000, 3),
                (1 << 23, 1_600_000, 3)]:
    rng = np.random.default_rng(0)
    colind = rng.integers(0, N, size=M * s).astype(np.int32)
    csrptr = (np.arange(M + 1, dtype=np.int64) * s).astype(np.int32)
    ptr, ind = torch.from_numpy(csrptr).to(dev), torch.from_numpy(colind).to(dev)
    plan = planmod.Plan.from_tensors(N, ptr, ind)
    X = torch.rand(N, F, device=dev)
    Xe = torch.empty(M, F, device=dev)
    ws = torch.empty(max(plan.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)

    def hop():
        _lib.check(_lib.lib().hg_gather_rows_f32(plan._h, 0, F, planmod._ptr(ptr), planmod._ptr(ind),
                   planmod._ptr(X), None, None, planmod._ptr(Xe), planmod._ptr(ws), ws.numel(),
                   planmod._stream_handle(X.device)))
    t = timeit(hop, 40)
    print(json.dumps({"N": N, "M": M, "size": s, "F": F, "table_MB": N * F * 4 / 1e6, "us": t * 1e6,
                      "Grows_per_s": M * s / t / 1e9, "gather_TBs": M * s * F * 4 / t / 1e12,
                      "write_TBs": M * F * 4 / t / 1e12}), flush=True)
    del plan
