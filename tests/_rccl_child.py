"""Child of tests/test_gpu_parity.py::test_rccl_branches_execute_at_world_size_one: ONE rank with the `nccl`
(= RCCL) backend on cuda:0.  A world of one needs no exchange, so the sharded aggregators are built with
force_collective=True: every collective of hypergef_amd.dist -- all-reduce, reduce_scatter_tensor, the
asynchronous column-pipelined forms, all_gather, the backward pass's all-reduce -- then goes through RCCL on the
GPU and is checked against the oracle (the checker; the per-rank operator is the product's HIP plan)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def main():
    out_dir = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl")  # before anything touches the GPU
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    from hypergef_amd import synth
    from hypergef_amd.dist import ColumnShardedAggregator, ShardedAggregator
    from oracle import oracle as orc
    orc.build()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    inc = synth.pubmed_shape()
    F = 32
    rng = np.random.default_rng(5)
    X = rng.standard_normal((inc.N, F)).astype(np.float32)
    W = (rng.random(inc.M) + 0.5).astype(np.float32)
    degE, degV = orc.degrees(inc.N, inc.M, inc.csrptr, inc.colind)
    Hp, Hi = orc.transpose_csr(inc.M, inc.N, inc.csrptr, inc.colind)
    ref = orc.hgnn_check(inc.N, inc.M, F, Hp, Hi, inc.csrptr, inc.colind, X, degE, degV, W)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Xd, dE, dV, Wd = t(X), t(degE), t(degV), t(W)
    tol = lambda y, r: np.abs(y - r) <= 1e-5 * np.maximum(1.0, np.abs(r))

    # count what actually reaches the process group: the test must not pass on skipped collectives
    calls = {"all_reduce": 0, "reduce_scatter_tensor": 0, "all_gather": 0}
    for name in calls:
        def wrap(fn, name=name):
            def inner(*a, **k):
                calls[name] += 1
                return fn(*a, **k)
            return inner
        setattr(dist, name, wrap(getattr(dist, name)))

    agg = ShardedAggregator(inc, device=dev, force_collective=True)
    assert agg._local_op == agg._hip_local_op
    Y = agg.aggregate(Xd, dE, dV, Wd)
    assert Y.is_cuda and tol(Y.cpu().numpy(), ref).all() and calls["all_reduce"] == 1
    rs = ShardedAggregator(inc, device=dev, exchange="reduce_scatter", force_collective=True)
    rows = rs.aggregate(Xd, dE, dV, Wd)
    assert rows.shape == (inc.N, F) and tol(rows.cpu().numpy(), ref).all() and calls["reduce_scatter_tensor"] == 1
    piped = ShardedAggregator(inc, device=dev, column_chunks=4, force_collective=True).aggregate(Xd, dE, dV, Wd)
    assert tol(piped.cpu().numpy(), ref).all() and calls["all_reduce"] == 5
    rows_p = ShardedAggregator(inc, device=dev, exchange="reduce_scatter", column_chunks=2,
                               force_collective=True).aggregate(Xd, dE, dV, Wd)
    assert tol(rows_p.cpu().numpy(), ref).all() and calls["reduce_scatter_tensor"] == 3
    cols = ColumnShardedAggregator(inc, device=dev, force_collective=True)
    assert tol(cols.aggregate(Xd, dE, dV, Wd, gather=True).cpu().numpy(), ref).all() and calls["all_gather"] == 1
    xg = Xd.clone().requires_grad_(True)
    G = rng.standard_normal((inc.N, F)).astype(np.float32)
    agg.apply(xg, dE, dV, Wd).backward(t(G))
    gref = orc.hgnn_check(inc.N, inc.M, F, Hp, Hi, inc.csrptr, inc.colind, G, degE, degV, W)
    assert tol(xg.grad.cpu().numpy(), gref).all() and calls["all_reduce"] == 7
    torch.cuda.synchronize()
    json.dump({"backend": dist.get_backend(), "world": dist.get_world_size(), "collective_calls": calls,
               "device": torch.cuda.get_device_name(0)}, open(os.path.join(out_dir, "rccl_world1.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
