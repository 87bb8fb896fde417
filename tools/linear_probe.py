#!/usr/bin/env python3
"""Aggregation with the layer's linear folded in (hg_aggr_linear_f32) against the two-step form
(torch linear = rocBLAS/hipBLASLt GEMM, then hg_aggr_fused_f32) on the bench shapes."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hypergef_amd import plan as planmod, synth
dev = "cuda:0"


def timed(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


rows = []
CONFIGS = (("cora", 1024, 32, 32), ("cora", 1024, 64, 64), ("cora", 256, 128, 128),
                              ("cora", 1024, 64, 16), ("citeseer", 1024, 64, 64), ("pubmed", 64, 128, 128),
                              ("pubmed", 64, 128, 64), ("cora", 1024, 32, 64), ("cora", 1024, 32, 48), ("cora", 1024, 64, 128))
for shape, K, F_in, F_out in CONFIGS[int(os.environ.get("PROBE_FROM", "0")):int(os.environ.get("PROBE_ONLY", "99"))]:
    inc = synth.replicate_block_diagonal(getattr(synth, shape + "_shape")(), K)
    ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
    X = torch.rand(inc.N, F_in, device=dev)
    Wl = torch.randn(F_out, F_in, device=dev) / F_in ** 0.5
    plan = planmod.Plan.from_tensors(inc.N, ptr, ind)
    Y = torch.empty(inc.N, F_out, device=dev)
    Z = torch.empty(inc.N, F_out, device=dev)
    T = torch.empty(inc.N, F_in, device=dev)
    ws = torch.empty(int(planmod._lib.lib().hg_aggr_linear_workspace_bytes(plan._h, F_in)) + 256,
                     dtype=torch.uint8, device=dev)
    ws2 = torch.empty(max(plan.workspace_bytes(max(F_in, F_out)), 256), dtype=torch.uint8, device=dev)
    plan.prepare(F_in), plan.prepare(F_out)
    wfrag = planmod.pack_linear(Wl)
    t_fused = timed(lambda: plan.aggregate_linear(ptr, ind, X, Wl, out=Y, workspace=ws, packed=wfrag))

    def two_step():
        torch.mm(X, Wl.t(), out=Z)
        plan.aggregate(ptr, ind, Z, out=Y, workspace=ws2)
    t_two = timed(two_step)
    t_gemm = timed(lambda: torch.mm(X, Wl.t(), out=Z))
    L = planmod._lib.lib()
    st = planmod._stream_handle(torch.device(dev))
    t_own = timed(lambda: planmod._lib.check(L.hg_linear_rows_f32(inc.N, F_in, F_out, X.data_ptr(), wfrag.data_ptr(),
                                                                   Z.data_ptr(), st)))
    t_agg_in = timed(lambda: plan.aggregate(ptr, ind, X, out=T, workspace=ws2))
    flops = 2.0 * inc.N * F_in * F_out
    bytes_min = 4.0 * inc.N * (F_in + F_out)
    rows.append({"shape": "%s x%d" % (shape, K), "F_in": F_in, "F_out": F_out, "fused_ms": t_fused,
                 "two_step_ms": t_two, "gemm_ms": t_gemm, "own_gemm_ms": t_own, "aggr_only_Fin_ms": t_agg_in,
                 "speedup": t_two / t_fused, "mfma_tflops": flops / t_fused / 1e9,
                 "hbm_frac_of_8TBs": bytes_min / t_fused / 1e6 / 8000.0})
    print(json.dumps(rows[-1]), flush=True)
