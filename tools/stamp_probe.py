#!/usr/bin/env python3
"""Per-phase cycle shares of fused_panel_kernel (diagnostic build path: HG_FUSED_DEBUG=32)."""
import ctypes, os, sys
os.environ["HG_FUSED_DEBUG"] = os.environ.get("STAMP_DEBUG", "32")
ROOT_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HG_AGGR_LIB", os.path.join(ROOT_, "hypergef_amd", "lib", "libhgaggr_stamps.so"))  # `make stamps`
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from hypergef_amd import plan as planmod, synth, _lib
dev = "cuda:0"
K, F = int(os.environ.get("STAMP_K", "1024")), int(os.environ.get("STAMP_F", "32"))
inc = synth.powerlaw(1_000_000, 4_000_000, seed=3) if os.environ.get("STAMP_SHAPE") == "powerlaw" else synth.replicate_block_diagonal(synth.cora_shape(), K)
ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
X = torch.rand(inc.N, F, device=dev)
plan = planmod.Plan.from_tensors(inc.N, ptr, ind)
Y = torch.empty(inc.N, F, device=dev)
ws = torch.empty(max(plan.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)
L = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    plan.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="fused")
torch.cuda.synchronize()
L.hg_debug_read_stamps(buf, 1)
n = 10
for _ in range(n):
    plan.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="fused")
torch.cuda.synchronize()
L.hg_debug_read_stamps(buf, 1)
names = os.environ.get("STAMP_NAMES", "descriptor,record copy (+scale reads),barrier 1,hop 1 (gather+tile),barrier 2,hop 2 (tile->Y)").split(",")
tot = sum(buf[i] for i in range(6))
info = plan.prepare(F)
waves = info["panels"] * 4 * n
clk = 1e8  # nominal: the tick is the shader clock here, so the us column is only relative
for i, nm in enumerate(names):
    print("%-28s %6.1f %%   %8.3f us/wave" % (nm, 100.0 * buf[i] / tot, buf[i] / waves / clk * 1e6))
print("F=%d panels %d cap %d: total us/wave %.3f" % (F, info["panels"], info["cap"], tot / waves / clk * 1e6))
