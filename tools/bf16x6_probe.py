#!/usr/bin/env python3
"""A/B of the linear epilogue's two matrix-phase forms at F_in = 128 (Options.linear_math): fp32 MFMA against six bf16
products per fp32 product.  Per shape: ms per call (alternating rounds, same box) and the error of both against a float64
product of the float64 aggregation, as max over elements of |got - ref| / max(1, row mass) (row mass = (|A| |X| |W^T|)[row]
summed: the bound the GPU tests use).
usage: tools/bf16x6_probe.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import scipy.sparse as sp
from hypergef_amd import plan as planmod, synth, _lib

dev = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
CASES = [("pubmed", 64, 128, 128), ("pubmed", 64, 128, 64), ("cora", 256, 128, 128), ("citeseer", 256, 128, 128),
         ("pubmed", 64, 128, 32), ("pubmed", 64, 128, 16)]
if os.environ.get("PROBE_CASES"):  # e.g. 0,2
    CASES = [CASES[int(i)] for i in os.environ["PROBE_CASES"].split(",")]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for shape, reps, F, Fo in CASES:
    base = getattr(synth, shape + "_shape")()
    inc = synth.replicate_block_diagonal(base, reps) if reps > 1 else base
    X_host = synth.features_like_reference(inc.N, F, seed=3)
    # signed features as well: cancellation is where a sloppy product would show
    X_host = (X_host - X_host.mean()).astype(np.float32)
    ptr = torch.from_numpy(inc.csrptr).to(dev)
    ind = torch.from_numpy(inc.colind).to(dev)
    X = torch.from_numpy(X_host).to(dev)
    g = torch.Generator().manual_seed(7)
    w_host = (torch.randn(Fo, F, generator=g) / F ** 0.5).numpy()
    weight = torch.from_numpy(w_host).to(dev)
    plan = planmod.Plan.from_tensors(inc.N, ptr, ind, planmod.make_opts())
    packed = planmod.pack_linear(weight, bf16x6=True)
    Y = {m: torch.empty((inc.N, Fo), dtype=torch.float32, device=dev) for m in ("f32", "bf16x6")}
    ws = torch.empty(max(int(_lib.lib().hg_aggr_linear_workspace_bytes(plan._h, F)), 256), dtype=torch.uint8, device=dev)
    T = torch.empty((inc.N, F), dtype=torch.float32, device=dev)

    RES = os.environ.get("PROBE_RES") == "1"  # a whole layer: t = 0.9 Aggr(X) + 0.1 R, relu (timing; the float64 columns then do not apply)
    R = torch.randn(inc.N, F, device=dev) if RES else None

    def run(m, t_out=None):
        plan.aggregate_linear(ptr, ind, X, weight, variant="fused", out=Y[m], workspace=ws, packed=packed, math=m, t_out=t_out,
                              residual=R, ca=0.9 if RES else 1.0, cb=0.1 if RES else 0.0, relu=RES)

    ms = {"f32": [], "bf16x6": []}
    for _ in range(rounds):
        for m in ("f32", "bf16x6"):
            ms[m].append(timeit(lambda: run(m)))
    # float64 truth: A = H H^T (vertex x vertex path counts), ref = (A X) W^T, mass = (A |X|) |W^T|
    H = sp.csr_matrix((np.ones(inc.nnz), inc.colind, inc.csrptr), shape=(inc.M, inc.N))  # hyperedge x vertex
    A = (H.T @ H).tocsr()
    AX = A @ X_host.astype(np.float64)
    ref = AX @ w_host.T.astype(np.float64)
    mass = (A @ np.abs(X_host).astype(np.float64)) @ np.abs(w_host.T).astype(np.float64)
    rowmass = np.maximum(1.0, mass.max(axis=1, keepdims=True))
    line = "%-9s x%-4d %3d->%-3d" % (shape, reps, F, Fo)
    for m in ("f32", "bf16x6"):
        run(m, T)
        torch.cuda.synchronize()
        got = Y[m].cpu().numpy().astype(np.float64)
        err_row = (np.abs(got - ref) / rowmass).max()
        err_el = (np.abs(got - ref) / np.maximum(1.0, mass)).max()
        t_ok = np.array_equal(T.cpu().numpy(), Y_T) if m == "bf16x6" else True
        if m == "f32":
            Y_T = T.cpu().numpy().copy()
        line += " | %s %.4f-%.4f ms err/rowmass %.2e err/elmass %.2e%s" % (m, min(ms[m]), max(ms[m]), err_row, err_el,
                                                                          "" if t_ok else " T_OUT DIFFERS")
    d = (Y["f32"] - Y["bf16x6"]).abs().max().item()
    print(line + " | max |f32 - bf16x6| %.3e" % d, flush=True)
