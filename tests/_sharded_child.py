"""Child of tests/test_gpu_parity.py::test_sharded_aggregator_two_ranks_share_the_gpu: one rank of a
world-2 run under torch.distributed.run, every rank on cuda:0, collectives over gloo.  The per-rank
operator is the product's default (HIP plan on the hyperedge shard); the oracle is the checker."""
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
    dist.init_process_group("gloo")  # before anything touches the GPU
    rank, world = dist.get_rank(), dist.get_world_size()
    from hypergef_amd import synth
    from hypergef_amd.dist import ColumnShardedAggregator, ShardedAggregator
    from oracle import oracle as orc
    orc.build()
    dev = torch.device("cuda", 0)
    inc = synth.pubmed_shape()
    F = 32
    rng = np.random.default_rng(3)
    X = rng.standard_normal((inc.N, F)).astype(np.float32)
    W = (rng.random(inc.M) + 0.5).astype(np.float32)
    degE, degV = orc.degrees(inc.N, inc.M, inc.csrptr, inc.colind)
    Hp, Hi = orc.transpose_csr(inc.M, inc.N, inc.csrptr, inc.colind)
    ref = orc.hgnn_check(inc.N, inc.M, F, Hp, Hi, inc.csrptr, inc.colind, X, degE, degV, W)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Xd, dE, dV, Wd = t(X), t(degE), t(degV), t(W)
    tol = lambda y, r: np.abs(y - r) <= 1e-5 * np.maximum(1.0, np.abs(r))

    agg = ShardedAggregator(inc, device=dev)  # exchange="allreduce", default (HIP) local operator
    assert (agg.rank, agg.world) == (rank, world) and agg._local_op == agg._hip_local_op
    Y = agg.aggregate(Xd, dE, dV, Wd)
    assert Y.is_cuda and tol(Y.cpu().numpy(), ref).all()
    both = [torch.empty_like(Y) for _ in range(world)]
    dist.all_gather(both, Y)
    assert all(torch.equal(b, both[0]) for b in both), "every rank holds the same sum"

    rs = ShardedAggregator(inc, device=dev, exchange="reduce_scatter")
    rows = rs.aggregate(Xd, dE, dV, Wd)
    lo, hi = rs.row_range()
    assert rows.shape == (hi - lo, F) and tol(rows.cpu().numpy(), ref[lo:hi]).all()

    # column slices pipelined against their collectives (SURVEY 8(e) iv), and column sharding (v): HIP operator
    piped = ShardedAggregator(inc, device=dev, column_chunks=4).aggregate(Xd, dE, dV, Wd)
    assert tol(piped.cpu().numpy(), ref).all()
    rows_p = ShardedAggregator(inc, device=dev, exchange="reduce_scatter", column_chunks=2).aggregate(Xd, dE, dV, Wd)
    assert rows_p.shape == (hi - lo, F) and tol(rows_p.cpu().numpy(), ref[lo:hi]).all()
    cols = ColumnShardedAggregator(inc, device=dev)
    c0, c1 = cols.columns(F)
    assert tol(cols.aggregate(Xd, dE, dV, Wd).cpu().numpy(), ref[:, c0:c1]).all()
    assert tol(cols.aggregate(Xd, dE, dV, Wd, gather=True).cpu().numpy(), ref).all()

    # autograd through the sharded operator: the reference's backward rule (forward on grad_out,
    # hgnnaggr.cc:51-64) applied shard by shard, partial gradients summed by the same collective
    xg = Xd.clone().requires_grad_(True)
    G = rng.standard_normal((inc.N, F)).astype(np.float32)
    agg.apply(xg, dE, dV, Wd).backward(t(G))
    gref = orc.hgnn_check(inc.N, inc.M, F, Hp, Hi, inc.csrptr, inc.colind, G, degE, degV, W)
    assert tol(xg.grad.cpu().numpy(), gref).all()

    torch.cuda.synchronize()
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
