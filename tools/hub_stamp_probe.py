#!/usr/bin/env python3
"""Per-phase tick shares of hub_pass_kernel on the power-law config (diagnostic build `make stamps`, HG_HUB_DEBUG=32)."""
import ctypes, os, sys
os.environ["HG_HUB_DEBUG"] = str(32 | int(os.environ.get("HUB_ABLATE", "0")))  # 1: no hop 1, 2: no hop 2, 4: no memory
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HG_AGGR_LIB", os.path.join(ROOT, "hypergef_amd", "lib", "libhgaggr_stamps.so"))
sys.path.insert(0, ROOT)
import torch
from hypergef_amd import plan as planmod, synth, _lib
dev = "cuda:0"
F = int(os.environ.get("STAMP_F", "64"))
inc = synth.powerlaw(1_000_000, 4_000_000, seed=3)
ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
X = torch.rand(inc.N, F, device=dev)
plan = planmod.Plan.from_tensors(inc.N, ptr, ind)
info = plan.prepare(F)
Y = torch.empty(inc.N, F, device=dev)
ws = torch.empty(max(plan.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)
L = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    plan.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="fused")
torch.cuda.synchronize()
L.hg_debug_read_stamps(buf, 1)
n = 5
for _ in range(n):
    plan.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="fused")
torch.cuda.synchronize()
L.hg_debug_read_stamps(buf, 1)
names = ["prologue (first record)", "round start: next-record fetch issue, header", "hop 1", "record stash + barrier", "hop 2", "barrier"]
tot = sum(buf[i] for i in range(6))
waves = info["hub_workgroups"] * 16
per_round = info["hub_rounds"] / info["hub_workgroups"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    plan.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="fused")
e1.record(); torch.cuda.synchronize()
print("ablate %s: step %.3f ms" % (os.environ.get("HUB_ABLATE", "0"), e0.elapsed_time(e1) / n))
for i, nm in enumerate(names):
    print("%-46s %5.1f %%   %8.1f ticks per wave and round" % (nm, 100.0 * buf[i] / tot, buf[i] / n / waves / per_round))
print("rounds %d workgroups %d entries %d pairs %d" % (info["hub_rounds"], info["hub_workgroups"], info["hub_entries"], info["hub_pairs"]))
