import sys, torch
sys.path.insert(0, '.')
from hypergef_amd import synth, plan as planmod
dev='cuda:0'
def t_us(f, n=20):
    for _ in range(10): f()
    g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/20/n*1e3
for name in (sys.argv[1:] or list(synth.ALLSET_SHAPES)):
    inc = synth.allset_shape(name)
    ptr, ind = torch.from_numpy(inc.csrptr).to(dev), torch.from_numpy(inc.colind).to(dev)
    import numpy as np
    sz = np.diff(inc.csrptr); print(name, 'N',inc.N,'M',inc.M,'nnz',inc.nnz,'max|e|',sz.max(),'mean',sz.mean())
    for F in (32, 64, 128):
        X = torch.rand(inc.N, F, device=dev); Y = torch.empty_like(X)
        for tb in (0,):
            pl = planmod.Plan.from_tensors(inc.N, ptr, ind, planmod.make_opts(t_big=tb))
            info = pl.prepare(F)  # before sizing the workspace: the fused schedule's partial rows count
            ws = torch.empty(max(pl.workspace_bytes(F),256), dtype=torch.uint8, device=dev)
            res = {}
            for v in ('auto','pull','fused'):
                res[v] = t_us(lambda: pl.aggregate(ptr, ind, X, out=Y, workspace=ws, variant=v))
            print(' F',F,'t_big',tb, pl.auto_variant(F), {k: round(x,2) for k,x in res.items()}, {k:info[k] for k in ('cap','panels','n_mat','n_hub','n_split','member_entries')})
