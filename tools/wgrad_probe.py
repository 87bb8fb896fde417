#!/usr/bin/env python3
"""hg_linear_wgrad_f32 against torch (rocBLAS) on the weight-gradient products of the layers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hypergef_amd.plan import linear_wgrad
dev = "cuda:0"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for N, Fa, Fb in ((693248, 64, 64), (2772992, 32, 32), (2772992, 64, 64), (693248, 32, 128), (1261888, 64, 64),
                  (693248, 128, 128), (1261888, 128, 128), (693248, 128, 64), (693248, 256, 128)):
    A = torch.randn(N, Fa, device=dev)
    B = torch.randn(N, Fb, device=dev)
    t_own = timed(lambda: linear_wgrad(A, B))
    t_torch = timed(lambda: A.t() @ B)
    floor = (A.numel() + B.numel()) * 4 / 5.4e12 * 1e3
    print("N %8d  %3d x %3d   own %.4f ms   torch %.4f ms   (%.1fx)   read-once floor %.4f ms" % (
        N, Fa, Fb, t_own, t_torch, t_torch / t_own, floor), flush=True)
