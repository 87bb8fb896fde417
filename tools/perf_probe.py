#!/usr/bin/env python3
"""Kernel-level probe: times hop 1, hop 2 and the full aggregation over a sweep
of batch sizes (working-set residency) for the current HG_* tuning env."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from hypergef_amd import plan as planmod, synth


def timeit(fn, iters, graph=True):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if graph:  # replay a captured batch of launches: no host overhead between kernels
        reps = max(1, min(iters, 50))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        outer = max(1, iters // reps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(outer):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (outer * reps) * 1e-3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="cora")
    ap.add_argument("--feat", type=int, default=32)
    ap.add_argument("--replicas", default="1024")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--panel-rows", type=int, default=0)
    ap.add_argument("--panel-nnz", type=int, default=0)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    dev = "cuda:0"
    base = {"cora": synth.cora_shape, "citeseer": synth.citeseer_shape, "pubmed": synth.pubmed_shape}[a.shape]()
    F = a.feat
    for K in [int(k) for k in a.replicas.split(",")]:
        inc = synth.replicate_block_diagonal(base, K)
        ptr = torch.from_numpy(inc.csrptr).to(dev)
        ind = torch.from_numpy(inc.colind).to(dev)
        X = torch.rand(inc.N, F, device=dev)
        plan = planmod.Plan.from_tensors(inc.N, ptr, ind,
                                         planmod.make_opts(panel_rows=a.panel_rows, panel_nnz=a.panel_nnz))
        Y = torch.empty(inc.N, F, device=dev)
        ws = torch.empty(max(plan.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)
        Xe = plan.gather_rows(0, ptr, ind, X)
        it = max(a.iters, min(2000, int(a.iters * 1024 / K)))
        Xe_out = torch.empty(inc.M, F, device=dev)
        from hypergef_amd import _lib
        import ctypes
        def hop(h, src, dst):
            _lib.check(_lib.lib().hg_gather_rows_f32(plan._h, h, F, planmod._ptr(ptr), planmod._ptr(ind),
                       planmod._ptr(src), None, None, planmod._ptr(dst), planmod._ptr(ws), ws.numel(),
                       planmod._stream_handle(X.device)))
        t1 = timeit(lambda: hop(0, X, Xe_out), it)
        t2 = timeit(lambda: hop(1, Xe, Y), it)
        tf = timeit(lambda: plan.aggregate(ptr, ind, X, out=Y, workspace=ws), it)
        NF, MF, nz = inc.N * F * 4, inc.M * F * 4, inc.nnz * 4
        b1 = NF + MF + nz + inc.M * 4      # read X, write Xe, indices, ptr
        b2 = MF + NF + nz + inc.N * 4      # read Xe, write Y, indices, ptr
        balg = 4 * (2 * inc.N * F + 2 * inc.nnz + inc.M + inc.N + 2 + inc.N)
        print(json.dumps({"tag": a.tag, "env": {k: v for k, v in os.environ.items() if k.startswith("HG_")},
                          "shape": a.shape, "F": F, "K": K, "MB_total": (b1 + b2) / 1e6,
                          "hop1_us": t1 * 1e6, "hop2_us": t2 * 1e6, "full_us": tf * 1e6,
                          "hop1_TBs": b1 / t1 / 1e12, "hop2_TBs": b2 / t2 / 1e12,
                          "full_TBs_actual": (b1 + b2) / tf / 1e12, "frac_alg": balg / tf / 8e12,
                          "us_per_block": tf * 1e6 / K}), flush=True)


if __name__ == "__main__":
    main()
