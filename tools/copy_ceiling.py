#!/usr/bin/env python3
"""What a plain device copy of the bench's X -> Y volume reaches on this box (read + write,
no gather): the practical HBM ceiling the fused kernel's 2NF-dominated traffic compares to."""
import sys, torch
dev = "cuda:0"
for rows in (2708 * 256, 2708 * 1024, 2708 * 2048, 2708 * 4096):
    X = torch.rand(rows, 32, device=dev)
    Y = torch.empty_like(X)
    for _ in range(5):
        Y.copy_(X)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    s.record()
    for _ in range(n):
        Y.copy_(X)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print("rows %9d  %7.1f MB moved  %.4f ms  %.2f TB/s" % (rows, 2 * X.numel() * 4 / 1e6, ms, 2 * X.numel() * 4 / ms / 1e9))
    # read-only (sum) and write-only (fill)
    s.record()
    for _ in range(n):
        Y.fill_(1.0)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print("   fill        %7.1f MB        %.4f ms  %.2f TB/s" % (X.numel() * 4 / 1e6, ms, X.numel() * 4 / ms / 1e9))
