"""Seeded synthetic incidence matrices in the shapes BASELINE.json names.

The reference's datasets are downloaded (HyperGsys/data/prepare.sh) and are not
available offline; SURVEY.md section 8(d) fixes these generators instead.  All
functions return the hypergraph as H_T in CSR (row = hyperedge, entries =
member vertices ascending), the layout the reference's operators consume
(HyperGsys/hypergraph.py:63-70).
"""
import numpy as np


class Incidence:
    """Host-side H_T CSR: csrptr int32 [M+1], colind int32 [nnz]."""

    def __init__(self, num_nodes, num_edges, csrptr, colind, name="synthetic"):
        self.N = int(num_nodes)
        self.M = int(num_edges)
        self.csrptr = np.ascontiguousarray(csrptr, dtype=np.int32)
        self.colind = np.ascontiguousarray(colind, dtype=np.int32)
        self.nnz = int(self.colind.shape[0])
        self.name = name
        assert self.csrptr.shape[0] == self.M + 1 and int(self.csrptr[-1]) == self.nnz

    def sizes(self):
        return np.diff(self.csrptr)


def _from_sizes(rng, N, sizes, name="synthetic"):
    """Members of each hyperedge drawn uniformly without replacement, ascending."""
    M = len(sizes)
    sizes = np.minimum(np.asarray(sizes, np.int64), N)
    csrptr = np.zeros(M + 1, np.int64)
    np.cumsum(sizes, out=csrptr[1:])
    colind = np.empty(int(csrptr[-1]), np.int32)
    for e in range(M):
        s = int(sizes[e])
        if s * 4 < N:
            mem = np.unique(rng.integers(0, N, size=s))
            while mem.shape[0] < s:  # top up after de-duplication
                mem = np.unique(np.concatenate([mem, rng.integers(0, N, size=s - mem.shape[0])]))
        else:
            mem = np.sort(rng.choice(N, size=s, replace=False))
        colind[csrptr[e]:csrptr[e + 1]] = mem
    return Incidence(N, M, csrptr.astype(np.int32), colind, name)


def cora_shape(seed=0):
    """C1/C2: N=2708, M=1579, |e| in {2..5} with p=(.35,.35,.2,.1)."""
    rng = np.random.default_rng(seed)
    sizes = rng.choice([2, 3, 4, 5], size=1579, p=[0.35, 0.35, 0.2, 0.1])
    return _from_sizes(rng, 2708, sizes, name="cora-shape")


def citeseer_shape(seed=1):
    """C2: N=3312, M=1079, mean |e| about 3.2, max 26."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.floor(2.0 * rng.random(1079) ** (-1.0 / 2.2)), 2, 26).astype(np.int64)
    return _from_sizes(rng, 3312, sizes, name="citeseer-shape")


def pubmed_shape(seed=2):
    """C3: N=19717, M=7963, mean |e| about 4.35, max 171 (truncated power law)."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.floor(2.0 * rng.random(7963) ** (-1.0 / 1.62)), 2, 171).astype(np.int64)
    return _from_sizes(rng, 19717, sizes, name="pubmed-shape")


def powerlaw(num_nodes=1_000_000, num_edges=4_000_000, seed=3, max_size=4096,
             tail=1.5, zipf=1.1):
    """C4: |e| = min(max_size, floor(2 u^(-1/tail))), members by Zipf(zipf)
    vertex popularity, de-duplicated per hyperedge.  Vectorised (no per-edge
    Python loop) so the 1M/4M case builds in seconds."""
    rng = np.random.default_rng(seed)
    sizes = np.minimum(max_size, np.floor(2.0 * rng.random(num_edges) ** (-1.0 / tail))).astype(np.int64)
    sizes = np.minimum(sizes, num_nodes)
    total = int(sizes.sum())
    pop = np.arange(1, num_nodes + 1, dtype=np.float64) ** (-zipf)
    cdf = np.cumsum(pop)
    cdf /= cdf[-1]
    # popularity rank -> vertex id through a fixed permutation so hubs are not ids 0..k
    perm = rng.permutation(num_nodes).astype(np.int64)
    eid = np.repeat(np.arange(num_edges, dtype=np.int64), sizes)
    mem = perm[np.searchsorted(cdf, rng.random(total))]
    key = np.unique(eid * num_nodes + mem)  # sort by (hyperedge, vertex) + dedupe
    eid, mem = key // num_nodes, key % num_nodes
    csrptr = np.zeros(num_edges + 1, np.int64)
    np.add.at(csrptr, eid + 1, 1)
    csrptr = np.cumsum(csrptr)
    return Incidence(num_nodes, num_edges, csrptr.astype(np.int32), mem.astype(np.int32),
                     name="powerlaw-%dx%d" % (num_nodes, num_edges))


def replicate_block_diagonal(inc, K):
    """K-fold block-diagonal replica (C2xK / C3xK of SURVEY.md 8(d)): a batch of
    K independent hypergraphs of the same shape, which is how the cora-scale
    shapes reach a working set beyond the 256 MiB Infinity Cache."""
    K = int(K)
    if K == 1:
        return inc
    nnz = inc.nnz
    colind = (inc.colind[None, :].astype(np.int64)
              + (np.arange(K, dtype=np.int64) * inc.N)[:, None]).reshape(-1)
    ptr = (inc.csrptr[None, :-1].astype(np.int64)
           + (np.arange(K, dtype=np.int64) * nnz)[:, None]).reshape(-1)
    ptr = np.concatenate([ptr, [K * nnz]])
    assert K * inc.N < 2 ** 31 and K * nnz < 2 ** 31
    return Incidence(inc.N * K, inc.M * K, ptr.astype(np.int32), colind.astype(np.int32),
                     name="%sx%d" % (inc.name, K))


def random_incidence(N, M, mean_size, seed=0, max_size=None, empty_frac=0.0):
    """Small ragged test matrices: geometric sizes, optional empty hyperedges."""
    rng = np.random.default_rng(seed)
    sizes = rng.geometric(1.0 / max(mean_size, 1.0), size=M).astype(np.int64)
    if max_size is not None:
        sizes = np.minimum(sizes, max_size)
    sizes = np.minimum(sizes, N)
    if empty_frac > 0:
        sizes[rng.random(M) < empty_frac] = 0
    return _from_sizes(rng, N, sizes, name="random")


def features_like_reference(n, F, seed=0):
    """U{0.0,...,0.9} as RamArray::fill_random_h draws them
    (include/util/ramArray.cuh:72-76), but from a seeded numpy stream."""
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 10, size=(n, F)).astype(np.float32) / np.float32(10)).astype(np.float32)


def write_mtx(path, inc):
    """`%%MatrixMarket matrix coordinate real general`, 1-based, value 1.0 --
    what scipy.io.mmwrite(H) emits for the reference (hypergraph.py:79-81).
    Rows are vertices, columns hyperedges (H, not H_T)."""
    e = np.repeat(np.arange(inc.M, dtype=np.int64), np.diff(inc.csrptr))
    v = inc.colind.astype(np.int64)
    order = np.lexsort((e, v))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%\n")
        f.write("%d %d %d\n" % (inc.N, inc.M, inc.nnz))
        for vi, ei in zip(v[order], e[order]):
            f.write("%d %d 1.0\n" % (vi + 1, ei + 1))


# Nominal AllSet / HyperGCN dataset statistics (vertices, hyperedges, incidences, largest
# hyperedge) for the 13 datasets of the reference's result table (experiment/example_data/
# result.xlsx, sheet "fig7,fig9").  The datasets themselves are downloaded by the reference
# (HyperGsys/data/prepare.sh) and are not available offline; these are the published shapes,
# quoted as nominal, used only to size synthetic stand-ins.
ALLSET_SHAPES = {
    "cora": (2708, 1579, 4859, 5),
    "citeseer": (3312, 1079, 3453, 26),
    "pubmed": (19717, 7963, 34629, 171),
    "coauthor_cora": (2708, 1072, 4585, 43),
    "coauthor_dblp": (41302, 22363, 99561, 202),
    "NTU2012": (2012, 2012, 10060, 5),
    "ModelNet40": (12311, 12311, 61555, 5),
    "zoo": (101, 43, 1717, 93),
    "Mushroom": (8124, 298, 40620, 1808),
    "20newsW100": (16242, 100, 65451, 2241),
    "house-committees": (1290, 341, 11843, 82),
    "walmart-trips": (88860, 69906, 460630, 25),
    "yelp": (50758, 679302, 2931130, 2838),
}


def allset_shape(name, seed=0):
    """Synthetic stand-in with the named dataset's nominal size: hyperedge sizes from a
    truncated power law fitted to (mean, max), members uniform without replacement."""
    N, M, nnz, smax = ALLSET_SHAPES[name]
    rng = np.random.default_rng(seed + sum(map(ord, name)))
    mean = nnz / M
    smin = 2 if mean >= 2.5 else 1
    smax = min(smax, N)
    if smax <= 6 or mean >= smax * 0.9:  # k-uniform (kNN-built) hypergraphs
        sizes = np.full(M, int(round(mean)), np.int64)
    else:
        lo, hi = 0.3, 6.0  # tail exponent: bisect until the mean matches
        u = rng.random(M)
        for _ in range(40):
            t = 0.5 * (lo + hi)
            sizes = np.clip(np.floor(smin * u ** (-1.0 / t)), smin, smax)
            if sizes.mean() > mean:
                lo = t
            else:
                hi = t
        sizes = sizes.astype(np.int64)
    total = int(sizes.sum())
    eid = np.repeat(np.arange(M, dtype=np.int64), sizes)
    mem = rng.integers(0, N, size=total).astype(np.int64)
    key = np.unique(eid * N + mem)  # de-duplicate inside each hyperedge (sizes shrink slightly)
    eid, mem = key // N, key % N
    csrptr = np.zeros(M + 1, np.int64)
    np.add.at(csrptr, eid + 1, 1)
    return Incidence(N, M, np.cumsum(csrptr).astype(np.int32), mem.astype(np.int32), name=name + "-shape")
