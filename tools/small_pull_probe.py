"""Single dataset-shaped hypergraphs: pull with the streaming row gather vs the panel / wave-task kernel, and fused
(device us per aggregation, 20 per hipGraph replay)."""
import sys, torch
sys.path.insert(0, '.')
from hypergef_amd import synth, plan as planmod
dev = 'cuda:0'


def t_us(f, n=20):
    for _ in range(10): f()
    g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 20 / n * 1e3


for name in (sys.argv[1:] or list(synth.ALLSET_SHAPES)):
    inc = synth.allset_shape(name)
    ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
    for F in (32, 64):
        X = torch.rand(inc.N, F, device=dev); Y = torch.empty_like(X)
        out = {}
        for label, opts in (("stream", planmod.make_opts()), ("panels", planmod.make_opts(row_stream=False))):
            pl = planmod.Plan.from_tensors(inc.N, ptr, ind, opts)
            pl.prepare(F)
            ws = torch.empty(max(pl.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)
            out["pull/" + label] = round(t_us(lambda: pl.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="pull")), 2)
            if label == "stream":
                out["fused"] = round(t_us(lambda: pl.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="fused")), 2)
                out["auto"] = pl.auto_variant(F)
                out["auto_us"] = round(t_us(lambda: pl.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="auto")), 2)
                tuned = pl.tune(ptr, ind, X)
                out["tuned"] = "%s/%d" % (tuned["variant"], tuned["pull_hop_kernels"])
                out["tuned_us"] = round(t_us(lambda: pl.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="auto")), 2)
        print("%-18s F=%-3d %s" % (name, F, out), flush=True)
