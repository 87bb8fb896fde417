"""MatrixMarket input with the reference DataLoader's semantics
(HyperGsys/include/dataloader/dataloader.hpp:22-180), read natively by
`hg_mtx_read` in libhgaggr."""
import ctypes

import numpy as np

from . import _lib
from .synth import Incidence


def read_mtx(path, name=None):
    """-> (Incidence of H_T, (H_ptr, H_ind)).  Rows of the file are vertices,
    columns hyperedges, exactly as `hyperg.store_mtx` writes them."""
    L = _lib.lib()
    i32p = ctypes.POINTER(ctypes.c_int32)
    nrow, ncol, nnz = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    ptrs = [i32p() for _ in range(4)]
    _lib.check(L.hg_mtx_read(str(path).encode(), ctypes.byref(nrow), ctypes.byref(ncol), ctypes.byref(nnz),
                             *[ctypes.byref(p) for p in ptrs]))
    try:
        N, M, nz = nrow.value, ncol.value, nnz.value
        H_ptr = np.ctypeslib.as_array(ptrs[0], shape=(N + 1,)).copy()
        H_ind = np.ctypeslib.as_array(ptrs[1], shape=(max(nz, 1),))[:nz].copy()
        HT_ptr = np.ctypeslib.as_array(ptrs[2], shape=(M + 1,)).copy()
        HT_ind = np.ctypeslib.as_array(ptrs[3], shape=(max(nz, 1),))[:nz].copy()
    finally:
        for p in ptrs:
            L.hg_free(p)
    return Incidence(N, M, HT_ptr, HT_ind, name or str(path)), (H_ptr, H_ind)
