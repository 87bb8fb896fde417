"""`balance_schedule`, the reference's edge-group partitioner, natively.

Same constructor and attributes as HyperGsys/balancer.py:4-33 (`balan_key`,
`balan_row`, `group_st`, `group_ed`, `nrow`, `ngs`, `work_p_sum`), computed by
`hg_balance_schedule` in libhgaggr (C++, O(M + sum w^2), int32 throughout)
instead of a Python loop with a `.item()` per row.  The attributes are int32
numpy arrays rather than Python lists; `torch.from_numpy` them directly (the
reference's `torch.Tensor(list).int()` goes through float32 and corrupts
indices above 2**24, HyperGsys/hypergraph.py:98-101).
"""
import ctypes

import numpy as np

from . import _lib


class balance_schedule:
    def __init__(self, ngs, H_T_csrptr):
        csrptr = np.ascontiguousarray(
            H_T_csrptr.detach().cpu().numpy() if hasattr(H_T_csrptr, "detach") else H_T_csrptr,
            dtype=np.int32)
        self.nrow = csrptr.shape[0] - 1
        self.ngs = int(ngs)
        self.balancer(csrptr)

    def balancer(self, csrptr):
        L = _lib.lib()
        nk, ng = ctypes.c_int64(0), ctypes.c_int64(0)
        p = csrptr.ctypes.data_as(ctypes.c_void_p)
        rc = L.hg_balance_schedule(self.nrow, self.ngs, p, ctypes.byref(nk), ctypes.byref(ng),
                                   None, None, None, None)
        if rc != _lib.HG_OK and self.nrow >= 0 and (csrptr.shape[0] == 0 or csrptr[-1] == 0):
            raise IndexError("list index out of range")  # what the reference raises on nnz == 0
        _lib.check(rc)
        key = np.empty(nk.value, np.int32)
        row = np.empty(ng.value, np.int32)
        st = np.empty(ng.value, np.int32)
        ed = np.empty(ng.value, np.int32)
        vp = ctypes.c_void_p
        _lib.check(L.hg_balance_schedule(self.nrow, self.ngs, p, ctypes.byref(nk), ctypes.byref(ng),
                                         key.ctypes.data_as(vp), row.ctypes.data_as(vp),
                                         st.ctypes.data_as(vp), ed.ctypes.data_as(vp)))
        self.balan_key, self.balan_row, self.group_st, self.group_ed = key, row, st, ed
        self.work_p_sum = int(key.shape[0] - 1)
