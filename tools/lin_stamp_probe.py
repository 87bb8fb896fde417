#!/usr/bin/env python3
"""Per-phase tick shares of the linear-epilogue instance of fused_packed_kernel (`make stamps`, HG_FUSED_DEBUG=32):
where a workgroup of hg_aggr_linear_f32 spends its life.  STAMP_SHAPE / STAMP_K / STAMP_F select the batch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HG_FUSED_DEBUG"] = os.environ.get("STAMP_DEBUG", "32")
os.environ.setdefault("HG_AGGR_LIB", os.path.join(ROOT, "hypergef_amd", "lib", "libhgaggr_stamps.so"))
sys.path.insert(0, ROOT)
import torch
from hypergef_amd import plan as planmod, synth, _lib
dev = "cuda:0"
shape, K, F = os.environ.get("STAMP_SHAPE", "pubmed"), int(os.environ.get("STAMP_K", "64")), int(os.environ.get("STAMP_F", "128"))
inc = synth.replicate_block_diagonal(getattr(synth, shape + "_shape")(), K)
ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
X = torch.rand(inc.N, F, device=dev)
Wl = torch.randn(F, F, device=dev) / F ** 0.5
plan = planmod.Plan.from_tensors(inc.N, ptr, ind)
Y = torch.empty(inc.N, F, device=dev)
L = _lib.lib()
ws = torch.empty(int(L.hg_aggr_linear_workspace_bytes(plan._h, F)) + 256, dtype=torch.uint8, device=dev)
wfrag = planmod.pack_linear(Wl, bf16x6=os.environ.get("STAMP_MATH") == "bf16x6")
buf = (ctypes.c_ulonglong * 16)()
plain = os.environ.get("STAMP_PLAIN") == "1"  # the plain aggregation's panels instead (phases 0-5, 12 only)
if plain:
    plan.prepare(F)
    ws = torch.empty(max(plan.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)
run = (lambda: plan.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="fused")) if plain else \
      (lambda: plan.aggregate_linear(ptr, ind, X, Wl, out=Y, workspace=ws, packed=wfrag, math=os.environ.get("STAMP_MATH", "f32")))
for _ in range(3):
    run()
torch.cuda.synchronize()
L.hg_debug_read_stamps(buf, 1)
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
L.hg_debug_read_stamps(buf, 1)
names = ["descriptor", "record copy issue (+scales)", "barrier 1 (record landed)", "hop 1 (gather + tile)", "barrier 2",
         "hop 2 -> registers, B prefetch issue", "barrier, rows -> LDS operand, barrier", "T_out", "matrix phase (MFMA loop)",
         "barrier after MFMA", "acc -> LDS, barrier", "row stores issue", "store acknowledgement (s_waitcnt vmcnt(0))"]
if plain:
    names[5] = "hop 2 (tile -> Y stores issue)"
tot = sum(buf[i] for i in range(13))
if buf[14]:
    print("shader clock while the sampled waves ran: %.2f GHz (s_memtime ticks / s_memrealtime ticks x 100 MHz)" % (buf[13] / buf[14] * 0.1))
out9 = (ctypes.c_int64 * 9)()
L.hg_debug_fused_shape.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
L.hg_debug_fused_shape(plan._h, F, 0 if plain else 1, out9)
waves = out9[0] * 4 * n
print("%s x%d F=%d: %.4f ms per call (with stamps); panels %d, rows/panel %.1f, cap %d" % (shape, K, F, e0.elapsed_time(e1) / n, out9[0], out9[1] / out9[0], out9[5]))
for i, nm in enumerate(names):
    i = 12 if i == 12 else i
    print("%-42s %6.1f %%   %9.1f ticks/wave" % (nm, 100.0 * buf[i] / tot, buf[i] / waves))
ms = e0.elapsed_time(e1) / n
per_cu_panel_us = ms * 1e3 * 256 / out9[0]
ghz = buf[13] / buf[14] * 0.1 if buf[14] else 2.4
life_us = tot / waves * 64 / (ghz * 1e3)  # one workgroup in 64 is sampled; s_memtime ticks are shader clocks
print("total ticks/wave %.1f -> %.1f us stamped per workgroup; the launch spends %.2f us per panel and CU: %.1f workgroups' worth of "
      "stamped time in flight per CU" % (tot / waves, life_us, per_cu_panel_us, life_us / per_cu_panel_us))
