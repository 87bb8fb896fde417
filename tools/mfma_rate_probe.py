#!/usr/bin/env python3
"""What the matrix pipes sustain on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, operands in registers) and the shader clock
under that load: the roofline's 157.3 TFLOP/s is 64 FLOP / clk / SIMD at 2.4 GHz.  hg_debug_mfma_rate (diagnostic entry)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hypergef_amd import _lib
torch.zeros(1, device="cuda:0")
L = _lib.lib()
L.hg_debug_mfma_rate.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
out = (ctypes.c_double * 3)()
for blocks, iters in ((2048, 2000), (2048, 20000), (2048, 100000), (1024, 100000)):
    rc = L.hg_debug_mfma_rate(blocks, iters, out)
    secs, tflops, ticks = out[0], out[1], out[2]
    print("blocks %d x 4 waves, %d x 16 MFMAs per wave: %.3f ms, %.1f TFLOP/s (%.2f of 157.3); one wave: %.0f s_memtime ticks = %.2f GHz if shader clocks"
          % (blocks, iters, secs * 1e3, tflops, tflops / 157.3, ticks, ticks / secs / 1e9), flush=True)
