import sys, torch
sys.path.insert(0, '.')
from hypergef_amd import synth, plan as planmod
dev='cuda:0'
def t_ms(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
for name, K in (("yelp",1),("yelp",4),("Mushroom",64),("20newsW100",64),("house-committees",256),("zoo",2048),("coauthor_dblp",32),("walmart-trips",8)):
    inc = synth.replicate_block_diagonal(synth.allset_shape(name), K)
    ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
    for F in (32, 64):
        X = torch.rand(inc.N, F, device=dev); Y = torch.empty_like(X)
        res = {}
        for rs in (True, False):
            pl = planmod.Plan.from_tensors(inc.N, ptr, ind, planmod.make_opts(row_stream=rs))
            pl.prepare(F)
            ws = torch.empty(max(pl.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)
            res[rs] = t_ms(lambda: pl.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="pull"))
            if rs: auto = pl.auto_variant(F); fused = t_ms(lambda: pl.aggregate(ptr, ind, X, out=Y, workspace=ws, variant="fused"))
        print("%-18s x%-5d F %3d max rows %s  pull stream %.4f  pull old %.4f  fused %.4f auto=%s" % (name, K, F, pl.info["max_len"], res[True], res[False], fused, auto), flush=True)
