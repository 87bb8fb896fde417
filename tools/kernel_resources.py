#!/usr/bin/env python3
"""Registers / scratch / occupancy per kernel instance from `make asm` (build/hg_kernels.resource.txt).
usage: tools/kernel_resources.py [substring of the demangled name ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, "build", "hg_kernels.resource.txt")).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for b, d in zip(blocks, dem):
    if sys.argv[1:] and not any(k in d for k in sys.argv[1:]):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print("%-110s VGPR %3s AGPR %3s scratch %4s occ %s" % (d.replace("void hg::", "")[:110], g("VGPRs"), g("AGPRs"),
          g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
