"""Multi-process paths on CPU (gloo, world_size 2; -m "not gpu").  The sharding and
the collective are the product code under test; the per-rank local operator is injected
(the oracle acts as checker-side stand-in for the HIP kernel, which needs a GPU)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from hypergef_amd import synth
from hypergef_amd.dist import (ColumnShardedAggregator, ShardedAggregator, local_incidence, partition_hyperedges,
                               shared_vertices)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_balances_incidences_and_covers():
    inc = synth.pubmed_shape()
    for world in (1, 2, 3, 8):
        parts = partition_hyperedges(inc.csrptr, world)
        assert parts[0][0] == 0 and parts[-1][1] == inc.M
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        loads = [int(inc.csrptr[hi] - inc.csrptr[lo]) for lo, hi in parts]
        assert sum(loads) == inc.nnz and max(loads) <= inc.nnz / world + inc.sizes().max()
    lo, hi = partition_hyperedges(inc.csrptr, 2)[1]
    loc = local_incidence(inc, lo, hi)
    assert loc.N == inc.N and loc.M == hi - lo
    assert np.array_equal(loc.colind, inc.colind[inc.csrptr[lo]:inc.csrptr[hi]])
    # a batch of independent hypergraphs sharded by graph shares no vertex
    batch = synth.replicate_block_diagonal(synth.cora_shape(), 4)
    cuts = [(i * 1579, (i + 1) * 1579) for i in range(4)]
    assert shared_vertices(batch, cuts).size == 0
    assert shared_vertices(inc, partition_hyperedges(inc.csrptr, 2)).size > 0


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    inc = synth.citeseer_shape()
    F = 6
    rng = np.random.default_rng(3)
    X = rng.standard_normal((inc.N, F)).astype(np.float32)
    W = (rng.random(inc.M) + 0.5).astype(np.float32)
    degE, degV = orc.degrees(inc.N, inc.M, inc.csrptr, inc.colind)

    def local_op(loc, Xt, dE, dV, Wt):
        Hp, Hi = orc.transpose_csr(loc.M, loc.N, loc.csrptr, loc.colind)
        y = orc.hgnn_check(loc.N, loc.M, Xt.shape[1], Hp, Hi, loc.csrptr, loc.colind, Xt.numpy(),
                           None if dE is None else dE.numpy(), None if dV is None else dV.numpy(),
                           None if Wt is None else Wt.numpy())
        return torch.from_numpy(y)

    agg = ShardedAggregator(inc, local_op=local_op)
    assert (agg.rank, agg.world) == (rank, world)
    Y = agg.aggregate(torch.from_numpy(X), torch.from_numpy(degE), torch.from_numpy(degV), torch.from_numpy(W))
    Hp, Hi = orc.transpose_csr(inc.M, inc.N, inc.csrptr, inc.colind)
    ref = orc.hgnn_check(inc.N, inc.M, F, Hp, Hi, inc.csrptr, inc.colind, X, degE, degV, W)
    np.testing.assert_allclose(Y.numpy(), ref, rtol=1e-5, atol=1e-6)
    # every rank holds the same full result
    gathered = [torch.empty_like(Y) for _ in range(world)]
    dist.all_gather(gathered, Y)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    # exchange="none": partials only, disjoint-support case adds up without a collective
    part = ShardedAggregator(inc, local_op=local_op, exchange="none").aggregate(
        torch.from_numpy(X), torch.from_numpy(degE), torch.from_numpy(degV), torch.from_numpy(W))
    tot = part.clone()
    dist.all_reduce(tot)
    np.testing.assert_allclose(tot.numpy(), ref, rtol=1e-5, atol=1e-6)
    # exchange="reduce_scatter": each rank ends with its block of rows of the sum
    rs = ShardedAggregator(inc, local_op=local_op, exchange="reduce_scatter")
    rows = rs.aggregate(torch.from_numpy(X), torch.from_numpy(degE), torch.from_numpy(degV), torch.from_numpy(W))
    lo, hi = rs.row_range()
    assert rows.shape == (hi - lo, F) and rs.row_range(world - 1)[1] == inc.N and rs.row_range(0)[0] == 0
    np.testing.assert_allclose(rows.numpy(), ref[lo:hi], rtol=1e-5, atol=1e-6)
    # column_chunks: the same sums, the collective of one column slice overlapping the next slice's aggregation
    args = (torch.from_numpy(X), torch.from_numpy(degE), torch.from_numpy(degV), torch.from_numpy(W))
    for chunks in (2, 3, 8):
        piped = ShardedAggregator(inc, local_op=local_op, column_chunks=chunks)
        assert [c1 - c0 for c0, c1 in piped.column_slices(F)] == {2: [3, 3], 3: [2, 2, 2], 8: [1] * 6}[chunks]
        assert torch.equal(piped.aggregate(*args), Y)
        rows_p = ShardedAggregator(inc, local_op=local_op, exchange="reduce_scatter", column_chunks=chunks).aggregate(*args)
        assert torch.equal(rows_p, rows)
    assert ShardedAggregator(inc, local_op=local_op, column_chunks=4).column_slices(64) == [(0, 16), (16, 32), (32, 48), (48, 64)]
    # column sharding: the whole hypergraph on every rank, F / world columns each, no collective
    cols = ColumnShardedAggregator(inc, local_op=local_op)
    c0, c1 = cols.columns(F)
    assert cols.columns(F, 0)[0] == 0 and cols.columns(F, world - 1)[1] == F
    np.testing.assert_allclose(cols.aggregate(*args).numpy(), ref[:, c0:c1], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cols.aggregate(*args, gather=True).numpy(), ref, rtol=1e-5, atol=1e-6)
    X5 = torch.from_numpy(np.ascontiguousarray(X[:, :5]))  # a width the ranks cannot split evenly
    ref5 = orc.hgnn_check(inc.N, inc.M, 5, Hp, Hi, inc.csrptr, inc.colind, X[:, :5].copy(), degE, degV, W)
    np.testing.assert_allclose(cols.aggregate(X5, *args[1:], gather=True).numpy(), ref5, rtol=1e-5, atol=1e-6)
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


def test_sharded_aggregation_allreduce_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def test_driver_data_parallel_gloo(tmp_path):
    """tools/hgsys.py under torchrun, 2 ranks, torch backend on CPU: DDP gradient all-reduce."""
    out = tmp_path / "o.csv"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tools", "hgsys.py"), "--backend", "torch", "--device", "cpu", "--epochs", "2",
           "--model", "HGNN", "--dname", "citeseer", "--replicas", "2", "--output", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "avg epoch time" in r.stdout and "2 rank(s)" in r.stdout
    assert out.read_text().startswith("torch,HGNN,citeseer,nlayer=2")
