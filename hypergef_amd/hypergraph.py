"""`HyperGraph`: the attribute bag the reference operators read.

Mirrors HyperGsys/hypergraph.py:10-101 for everything on the aggregation path:
`num_nodes, num_edges, nnz, degV [N,1], degE [M,1], H_csrptr/H_colind/H_data,
H_T_csrptr/H_T_colind/H_T_data, group_key/group_row/group_start/group_end`.
The DGL parts (`L`, `dgl_prepare`) are out of scope.
"""
import numpy as np
import torch

from .balancer import balance_schedule
from .synth import Incidence

# per-dataset partition sizes, HyperGsys/hypergraph.py:74-75
PARTITION_DICT = {"yelp": 400, "20newsW100": 400, "coauthor_cora": 10, "zoo": 20, "NTU2012": 80,
                  "cora": 210, "pubmed": 40, "Mushroom": 250, "coauthor_dblp": 80,
                  "house-committees": 40, "walmart-trips": 210, "citeseer": 6, "ModelNet40": 300}


def _transpose_csr(nrows, ncols, ptr, ind):
    """Stable counting-sort transpose (dataloader.hpp:121-141 semantics)."""
    order = np.argsort(ind, kind="stable")
    rows = np.repeat(np.arange(nrows, dtype=np.int32), np.diff(ptr))
    t_ind = rows[order].astype(np.int32)
    t_ptr = np.zeros(ncols + 1, np.int64)
    np.add.at(t_ptr, ind.astype(np.int64) + 1, 1)
    return np.cumsum(t_ptr).astype(np.int32), t_ind


class HyperGraph:
    def __init__(self, data, device, data_name, ngs=None):
        """`data` carries `.x` ([N, *]) and `.edge_index` ([2, 2*nnz] V2E then E2V,
        hyperedge ids offset by N), as the reference's PyG `Data` does
        (hypergraph.py:14-21).  Use `from_incidence` for a CSR you already hold."""
        num_nodes = data.x.shape[0]
        ei = torch.as_tensor(data.edge_index).cpu()
        c_idx = int(torch.where(ei[0] == num_nodes)[0].min())
        V = ei[0, :c_idx].numpy().astype(np.int64)
        E = ei[1, :c_idx].numpy().astype(np.int64) - num_nodes
        num_edges = int(np.unique(E).shape[0])
        order = np.lexsort((V, E))
        ptr = np.zeros(num_edges + 1, np.int64)
        np.add.at(ptr, E + 1, 1)
        inc = Incidence(num_nodes, num_edges, np.cumsum(ptr), V[order], data_name)
        self._init(inc, device, data_name, ngs)

    @classmethod
    def from_incidence(cls, inc, device, data_name=None, ngs=None):
        self = cls.__new__(cls)
        self._init(inc, device, data_name or inc.name, ngs)
        return self

    def _init(self, inc, device, data_name, ngs):
        self.device = torch.device(device)
        self.data_name = data_name
        self.num_nodes, self.num_edges, self.nnz = inc.N, inc.M, inc.nnz
        N, M = inc.N, inc.M
        HT_ptr, HT_ind = inc.csrptr, inc.colind
        H_ptr, H_ind = _transpose_csr(M, N, HT_ptr, HT_ind)

        # hypergraph.py:34-49: degV = rowsum^-1/2 (inf -> 1), degE = colsum^-1 (no guard)
        degV = torch.from_numpy(np.diff(H_ptr).astype(np.float32)).reshape(N, 1).pow(-0.5)
        degE = torch.from_numpy(np.diff(HT_ptr).astype(np.float32)).reshape(M, 1).pow(-1)
        degD = degV.pow(-1)
        degV[torch.isinf(degV)] = 1
        self.degV, self.degE, self.degD = degV.to(device), degE.to(device), degD.to(device)

        self.H_csrptr = torch.from_numpy(H_ptr).to(device)
        self.H_colind = torch.from_numpy(H_ind).to(device)
        self.H_data = torch.ones(inc.nnz, dtype=torch.float32, device=device)
        self.H_T_csrptr = torch.from_numpy(HT_ptr).to(device)
        self.H_T_colind = torch.from_numpy(HT_ind).to(device)
        self.H_T_data = torch.ones(inc.nnz, dtype=torch.float32, device=device)
        self.adj_g1 = self.H_csrptr, self.H_colind, self.H_data
        self.adj_g2 = self.H_T_csrptr, self.H_T_colind, self.H_T_data
        self._host = inc

        if ngs is None:
            ngs = PARTITION_DICT.get(data_name, 210)
        self.balance(ngs, HT_ptr)

    def balance(self, ngs, H_T_csrptr):
        bs = balance_schedule(ngs, H_T_csrptr)
        self.ngs = ngs
        dev = self.device
        self.group_start = torch.from_numpy(bs.group_st).to(dev)
        self.group_end = torch.from_numpy(bs.group_ed).to(dev)
        self.group_key = torch.from_numpy(bs.balan_key).to(dev)
        self.group_row = torch.from_numpy(bs.balan_row).to(dev)

    def store_mtx(self, path):
        from .synth import write_mtx
        write_mtx(path + self.data_name + ".mtx", self._host)
