"""Multi-GPU: one process per GPU, hypergraph sharded by hyperedge group.

The reference is single-GPU; this is new work defined by
BASELINE.json.  Each hyperedge's two hops need only X, so hyperedges are cut
into `world` contiguous groups balanced by incidence count; every rank holds X
and degV replicated, aggregates its own hyperedges into a dense partial
Y_r [N, F], and one sum all-reduce over RCCL (backend "nccl" on ROCm) yields Y
on every rank:  Y = sum_r degV . H_r (degE_r . W_r . (H_r^T X)).

`exchange="reduce_scatter"` (SURVEY.md 8(e) option ii) moves half the bytes: the sum is
scattered, rank r ends with rows `row_range(r)` of Y only -- what a row-parallel next layer needs.

When no vertex is shared between shards (a batch of independent hypergraphs
sharded by graph) the partials have disjoint row support; `exchange="none"`
then skips the collective and leaves Y row-sharded (each rank owns the rows of
its graphs), which is the weak-scaling form bench.py reports.

`column_chunks=k` (SURVEY.md 8(e) option iv) cuts the feature columns into k slices: slice c's collective runs
on the communication stream while slice c+1's kernels run, so the all-reduce hides the aggregation (or the
other way round) instead of following it.  `ColumnShardedAggregator` (option v) needs no collective at all:
every rank holds the whole hypergraph and owns F / world columns of X and Y.
"""
import numpy as np
import torch
import torch.distributed as dist

from .synth import Incidence


def partition_hyperedges(csrptr, world):
    """Contiguous hyperedge ranges [lo_r, hi_r) with near-equal incidence counts."""
    csrptr = np.asarray(csrptr, np.int64)
    M = csrptr.shape[0] - 1
    nnz = int(csrptr[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(csrptr, nnz * r / world, side="left")))
    cuts.append(M)
    cuts = np.maximum.accumulate(np.minimum(cuts, M))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def local_incidence(inc, lo, hi):
    """Rows [lo, hi) of H_T, vertex ids unchanged (X stays replicated)."""
    ptr = inc.csrptr[lo:hi + 1].astype(np.int64)
    return Incidence(inc.N, hi - lo, ptr - ptr[0], inc.colind[ptr[0]:ptr[-1]],
                     name="%s[%d:%d]" % (inc.name, lo, hi))


def shared_vertices(inc, parts):
    """Vertices touched by more than one shard (the rows a sparse exchange would move)."""
    owner_count = np.zeros(inc.N, np.int32)
    for lo, hi in parts:
        v = np.unique(inc.colind[inc.csrptr[lo]:inc.csrptr[hi]])
        owner_count[v] += 1
    return np.nonzero(owner_count > 1)[0]


class ShardedAggregator:
    """Per-rank object.  `local_op(inc_local, X, degE_local, degV, W_local) -> Y_partial`
    defaults to the HIP path; tests inject a CPU checker to exercise the sharding
    and the collective over gloo without a GPU."""

    def __init__(self, inc, rank=None, world=None, device=None, local_op=None, exchange="allreduce",
                 column_chunks=1, force_collective=False):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        # A world of one needs no exchange and skips it.  force_collective=True issues the collectives anyway
        # (a one-rank communicator): the RCCL code path of this class -- all-reduce, padded reduce-scatter, the
        # asynchronous pipelined form -- can then be executed and checked on a single GPU.
        self._collective = self.world > 1 or bool(force_collective)
        self.parts = partition_hyperedges(inc.csrptr, self.world)
        self.lo, self.hi = self.parts[self.rank]
        self.local = local_incidence(inc, self.lo, self.hi)
        self.N, self.M = inc.N, inc.M
        self.device = device
        if exchange not in ("allreduce", "reduce_scatter", "none"):
            raise ValueError("exchange must be 'allreduce', 'reduce_scatter' or 'none'")
        self.exchange = exchange
        if column_chunks < 1:
            raise ValueError("column_chunks must be at least 1")
        self.column_chunks = int(column_chunks)
        self._local_op = local_op or self._hip_local_op
        self._hip = None

    def _hip_local_op(self, inc_local, X, degE, degV, W):
        from .plan import Plan
        if self._hip is None:
            ptr = torch.from_numpy(inc_local.csrptr).to(X.device)
            ind = torch.from_numpy(inc_local.colind).to(X.device)
            self._hip = (Plan.from_tensors(self.N, ptr, ind), ptr, ind)
        plan, ptr, ind = self._hip
        return plan.aggregate(ptr, ind, X, degE, degV, W)

    def slice_edge_vector(self, t):
        """degE / W are per hyperedge: each rank uses its own slice."""
        return None if t is None else t.reshape(-1)[self.lo:self.hi].contiguous()

    def aggregate(self, X, degE=None, degV=None, W=None):
        dE, dV, dW = self.slice_edge_vector(degE), None if degV is None else degV.reshape(-1), self.slice_edge_vector(W)
        if self.column_chunks > 1 and self._collective and self.exchange != "none":
            return self._aggregate_pipelined(X, dE, dV, dW)
        Y = self._local_op(self.local, X, dE, dV, dW)
        if self.exchange == "allreduce" and self._collective:
            dist.all_reduce(Y, op=dist.ReduceOp.SUM)
        elif self.exchange == "reduce_scatter":
            return self._reduce_scatter(Y)
        return Y

    def column_slices(self, F):
        """[c0, c1) of every chunk: near-equal widths, multiples of 4 floats where F allows."""
        k = max(1, min(self.column_chunks, F))
        unit = 4 if F % 4 == 0 and F // 4 >= k else 1
        n = F // unit
        cuts = [(n * c // k) * unit for c in range(k)] + [F]
        return [(cuts[c], cuts[c + 1]) for c in range(k) if cuts[c + 1] > cuts[c]]

    def _aggregate_pipelined(self, X, dE, dV, dW):
        """Column slice c's collective is issued asynchronously (RCCL runs it on its own stream, ordered after
        the kernels that produced the slice) and overlaps slice c + 1's aggregation; the slices are put back
        together once their collectives have finished."""
        F = X.shape[1]
        lo, hi = self.row_range() if self.exchange == "reduce_scatter" else (0, self.N)
        out = X.new_empty((hi - lo, F))
        gloo = dist.get_backend() == "gloo"
        blk = -(-self.N // self.world)
        pending = []
        for c0, c1 in self.column_slices(F):
            Yc = self._local_op(self.local, X[:, c0:c1].contiguous(), dE, dV, dW)
            if self.exchange == "allreduce" or gloo:
                work, res = dist.all_reduce(Yc, op=dist.ReduceOp.SUM, async_op=True), Yc
            else:
                pad = blk * self.world - self.N
                src = Yc if pad == 0 else torch.cat([Yc, Yc.new_zeros((pad, c1 - c0))])
                res = Yc.new_empty((blk, c1 - c0))
                work = dist.reduce_scatter_tensor(res, src, op=dist.ReduceOp.SUM, async_op=True)
                pending.append((None, None, src, None))  # keep the source alive until the wait below
            pending.append((c0, c1, res, work))
        for c0, c1, res, work in pending:
            if work is None:
                continue
            work.wait()
            if self.exchange == "allreduce":
                out[:, c0:c1] = res
            elif gloo:
                out[:, c0:c1] = res[lo:hi]
            else:
                out[:, c0:c1] = res[:hi - lo]
        return out

    def apply(self, X, degE=None, degV=None, W=None):
        """`aggregate` as an autograd node (training): gradient for X only, as the reference's
        operators (hgnnaggr.cc:51-64).  Backward follows the calling thread's ops.Options.backward: "reference" = the
        forward operator on grad_out, "adjoint" = H diag(degE W) H^T diag(degV) grad; either way
        each rank applies its shard's operator to the (replicated) grad_out and the partial
        gradients are summed by the same collective, because X is replicated.  Needs a full Y
        on every rank, i.e. exchange "allreduce" (or world 1)."""
        if self.exchange == "reduce_scatter" and self.world > 1:
            raise ValueError("autograd through exchange='reduce_scatter' is not supported: grad_out would be row-sharded")
        return _ShardedFn.apply(self, X, degE, degV, W)

    def row_range(self, rank=None):
        """Rows of Y that `exchange="reduce_scatter"` leaves on `rank`: equal blocks of
        ceil(N / world) rows, the last one short."""
        rank = self.rank if rank is None else rank
        blk = -(-self.N // self.world)
        return min(rank * blk, self.N), min((rank + 1) * blk, self.N)

    def _reduce_scatter(self, Y):
        lo, hi = self.row_range()
        if not self._collective:
            return Y
        blk = -(-self.N // self.world)
        F = Y.shape[1]
        if dist.get_backend() == "gloo":  # no reduce_scatter in gloo (CPU tests): same result, more bytes
            dist.all_reduce(Y, op=dist.ReduceOp.SUM)
            return Y[lo:hi].contiguous()
        pad = blk * self.world - self.N
        src = Y if pad == 0 else torch.cat([Y, Y.new_zeros((pad, F))])
        out = Y.new_empty((blk, F))
        dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM)
        return out[:hi - lo]


class _ShardedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, agg, X, degE, degV, W):
        from . import ops
        ctx.agg = agg
        ctx.backward_mode = ops.current_options().backward  # the forward's options, whichever thread runs the backward
        ctx.save_for_backward(degE, degV, W)
        return agg.aggregate(X, degE, degV, W)

    @staticmethod
    def backward(ctx, grad_out):
        from . import ops
        degE, degV, W = ctx.saved_tensors
        g = grad_out.contiguous()
        if ctx.backward_mode == "reference" or degV is None:
            gx = ctx.agg.aggregate(g, degE, degV, W)
        else:
            gx = ctx.agg.aggregate(g * degV.reshape(-1, 1), degE, None, W)
        return None, gx, None, None, None


class ColumnShardedAggregator:
    """SURVEY.md 8(e) option v: no collective.  Every rank holds the whole hypergraph and owns a contiguous
    block of feature columns; `aggregate(X)` takes the full [N, F] X (or the rank's own [N, F_r] block with
    `sliced=True`) and returns Y[:, columns(rank)].  The narrower rows cost coalescing (F = 64 over 8 ranks is
    32-byte rows); `gather=True` all-gathers the blocks into the full Y on every rank."""

    def __init__(self, inc, rank=None, world=None, device=None, local_op=None, force_collective=False):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self._collective = self.world > 1 or bool(force_collective)  # as ShardedAggregator
        self.inc = inc
        self.N, self.M = inc.N, inc.M
        self.device = device
        self._local_op = local_op or self._hip_local_op
        self._hip = None

    def _hip_local_op(self, inc, X, degE, degV, W):
        from .plan import Plan
        if self._hip is None:
            ptr = torch.from_numpy(inc.csrptr).to(X.device)
            ind = torch.from_numpy(inc.colind).to(X.device)
            self._hip = (Plan.from_tensors(self.N, ptr, ind), ptr, ind)
        plan, ptr, ind = self._hip
        return plan.aggregate(ptr, ind, X, degE, degV, W)

    def columns(self, F, rank=None):
        """[c0, c1) of `rank`: blocks of ceil(F / world) columns, the last ones short or empty."""
        rank = self.rank if rank is None else rank
        blk = -(-F // self.world)
        return min(rank * blk, F), min((rank + 1) * blk, F)

    def aggregate(self, X, degE=None, degV=None, W=None, sliced=False, gather=False, F=None):
        F = X.shape[1] if not sliced else (F if F is not None else X.shape[1] * self.world)
        c0, c1 = self.columns(F)
        Xr = X if sliced else X[:, c0:c1].contiguous()
        flat = lambda t: None if t is None else t.reshape(-1)
        if c1 > c0:
            Yr = self._local_op(self.inc, Xr, flat(degE), flat(degV), flat(W))
        else:
            Yr = X.new_empty((self.N, 0))
        if not gather or not self._collective:
            return Yr
        blk = -(-F // self.world)
        mine = Yr if Yr.shape[1] == blk else torch.cat([Yr, Yr.new_zeros((self.N, blk - Yr.shape[1]))], dim=1)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine.contiguous())
        return torch.cat(parts, dim=1)[:, :F].contiguous()
