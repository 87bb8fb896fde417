"""CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of ``oracle/hg_oracle.c`` plus numpy restatements of the
pieces of the reference that are Python (degree vectors, MatrixMarket
semantics).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; ``hypergef_amd`` never does.

Pinning: ``balance_schedule`` is pinned by ``tests/golden/balancer_*.npz``
(generated from the reference's ``HyperGsys/balancer.py``).  The aggregation
functions are **parity unpinned** (no reference golden vectors exist and no
reference code for them can run in this image; see ``hg_oracle.c`` header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhg_oracle.so")
_lib = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile hg_oracle.c with the committed Makefile (gcc, seconds)."""
    src = os.path.join(_HERE, "hg_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.run(["make", "-C", _HERE, "-B", "libhg_oracle.so"],
                       check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_balance_schedule.restype = ctypes.c_int
        _lib.oracle_check_result.restype = ctypes.c_int
    return _lib


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def _f32(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    return a, a.ctypes.data_as(_f32p)


def transpose_csr(nrow, ncol, indptr, indices):
    """dataloader.hpp:107-141 (stable counting sort)."""
    indptr, p0 = _i32(indptr)
    indices, p1 = _i32(indices)
    t_ptr = np.zeros(ncol + 1, np.int32)
    t_ind = np.zeros(indices.shape[0], np.int32)
    lib().oracle_transpose_csr(ctypes.c_int32(nrow), ctypes.c_int32(ncol), p0, p1,
                               t_ptr.ctypes.data_as(_i32p), t_ind.ctypes.data_as(_i32p))
    return t_ptr, t_ind


def hyperaggr_host(N, F, H_ptr, H_ind, HT_ptr, HT_ind, X, omp=False):
    """check.cuh:83-114 -- unweighted Y = H H^T X, the reference CPU path."""
    H_ptr, a0 = _i32(H_ptr)
    H_ind, a1 = _i32(H_ind)
    HT_ptr, a2 = _i32(HT_ptr)
    HT_ind, a3 = _i32(HT_ind)
    X, a4 = _f32(X)
    out = np.empty(N * F, np.float32)
    fn = lib().oracle_hyperaggr_host_omp if omp else lib().oracle_hyperaggr_host
    fn(ctypes.c_int32(N), ctypes.c_int32(F), a0, a1, a2, a3, a4, out.ctypes.data_as(_f32p))
    return out.reshape(N, F)


def twostep_host(N, M, F, H_ptr, H_ind, HT_ptr, HT_ind, X):
    """spmm.cuh:724-740 -- two spmm_reference_host calls."""
    H_ptr, a0 = _i32(H_ptr)
    H_ind, a1 = _i32(H_ind)
    HT_ptr, a2 = _i32(HT_ptr)
    HT_ind, a3 = _i32(HT_ind)
    X, a4 = _f32(X)
    tmp = np.empty(M * F, np.float32)
    out = np.empty(N * F, np.float32)
    lib().oracle_twostep_host(ctypes.c_int32(N), ctypes.c_int32(M), ctypes.c_int32(F),
                              a0, a1, a2, a3, a4,
                              tmp.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p))
    return out.reshape(N, F), tmp.reshape(M, F)


def hgnn_check(N, M, F, H_ptr, H_ind, HT_ptr, HT_ind, X, degE=None, degV=None, W=None,
               return_xe=False):
    """test/hgnn_test.py:56-63 -- weighted operator, scale after each sum."""
    H_ptr, a0 = _i32(H_ptr)
    H_ind, a1 = _i32(H_ind)
    HT_ptr, a2 = _i32(HT_ptr)
    HT_ind, a3 = _i32(HT_ind)
    X, a4 = _f32(X)
    degE, a5 = _f32(degE)
    degV, a6 = _f32(degV)
    W, a7 = _f32(W)
    Xe = np.empty(M * F, np.float32)
    Y = np.empty(N * F, np.float32)
    lib().oracle_hgnn_check(ctypes.c_int32(N), ctypes.c_int32(M), ctypes.c_int32(F),
                            a0, a1, a2, a3, a4, a5, a6, a7,
                            Xe.ctypes.data_as(_f32p), Y.ctypes.data_as(_f32p))
    if return_xe:
        return Y.reshape(N, F), Xe.reshape(M, F)
    return Y.reshape(N, F)


def hgnn_kernel_order(N, M, F, HT_ptr, HT_ind, X, degE=None, degV=None, W=None):
    """hgnnaggr_cuda.cu:28-45 arithmetic with atomics replayed in task order."""
    HT_ptr, a2 = _i32(HT_ptr)
    HT_ind, a3 = _i32(HT_ind)
    X, a4 = _f32(X)
    degE, a5 = _f32(degE)
    degV, a6 = _f32(degV)
    W, a7 = _f32(W)
    Y = np.empty(N * F, np.float32)
    lib().oracle_hgnn_kernel_order(ctypes.c_int32(N), ctypes.c_int32(M), ctypes.c_int32(F),
                                   a2, a3, a4, a5, a6, a7, Y.ctypes.data_as(_f32p))
    return Y.reshape(N, F)


def hgnn_mean(N, M, F, HT_ptr, HT_ind, X, degE, degV, W):
    """hgnnaggr_cuda.cu:86-113 (loop bound fixed to M, defect D2)."""
    HT_ptr, a2 = _i32(HT_ptr)
    HT_ind, a3 = _i32(HT_ind)
    X, a4 = _f32(X)
    degE, a5 = _f32(degE)
    degV, a6 = _f32(degV)
    W, a7 = _f32(W)
    Y = np.empty(N * F, np.float32)
    lib().oracle_hgnn_mean(ctypes.c_int32(N), ctypes.c_int32(M), ctypes.c_int32(F),
                           a2, a3, a4, a5, a6, a7, Y.ctypes.data_as(_f32p))
    return Y.reshape(N, F)


def hgnn_max(N, M, F, HT_ptr, HT_ind, X, degE, degV, W):
    """hgnnaggr_cuda.cu:144-177 (loop bound fixed to M, defect D2)."""
    HT_ptr, a2 = _i32(HT_ptr)
    HT_ind, a3 = _i32(HT_ind)
    X, a4 = _f32(X)
    degE, a5 = _f32(degE)
    degV, a6 = _f32(degV)
    W, a7 = _f32(W)
    Y = np.empty(N * F, np.float32)
    rec = np.zeros(M * F, np.int32)
    lib().oracle_hgnn_max(ctypes.c_int32(N), ctypes.c_int32(M), ctypes.c_int32(F),
                          a2, a3, a4, a5, a6, a7, Y.ctypes.data_as(_f32p),
                          rec.ctypes.data_as(_i32p))
    return Y.reshape(N, F), rec.reshape(M, F)


def balance_schedule(ngs, csrptr):
    """balancer.py:15-33 -> (key, row, group_st, group_ed) int32 arrays."""
    csrptr, p = _i32(csrptr)
    nrow = csrptr.shape[0] - 1
    nk = ctypes.c_int64(0)
    ng = ctypes.c_int64(0)
    rc = lib().oracle_balance_schedule(ctypes.c_int32(nrow), ctypes.c_int32(ngs), p,
                                       ctypes.byref(nk), ctypes.byref(ng),
                                       None, None, None, None)
    if rc != 0:
        raise IndexError("balance_schedule on an empty incidence matrix (reference: list index out of range)")
    key = np.empty(nk.value, np.int32)
    row = np.empty(ng.value, np.int32)
    st = np.empty(ng.value, np.int32)
    ed = np.empty(ng.value, np.int32)
    lib().oracle_balance_schedule(ctypes.c_int32(nrow), ctypes.c_int32(ngs), p,
                                  ctypes.byref(nk), ctypes.byref(ng),
                                  key.ctypes.data_as(_i32p), row.ctypes.data_as(_i32p),
                                  st.ctypes.data_as(_i32p), ed.ctypes.data_as(_i32p))
    return key, row, st, ed


def check_result(C, C_ref):
    """check.cuh:40-57 relative-1e-2 verdict."""
    C = np.ascontiguousarray(C, np.float32)
    C_ref = np.ascontiguousarray(C_ref, np.float32)
    M, N = C_ref.shape
    return bool(lib().oracle_check_result(ctypes.c_int32(M), ctypes.c_int32(N),
                                          C.ctypes.data_as(_f32p), C_ref.ctypes.data_as(_f32p)))


def fill_random(n, seed=1):
    """ramArray.cuh:72-76 values (rand()%10)/10."""
    a = np.empty(n, np.float32)
    lib().oracle_fill_random(a.ctypes.data_as(_f32p), ctypes.c_int64(n), ctypes.c_uint(seed))
    return a


def num_threads():
    return int(lib().oracle_num_threads())


# ---------------------------------------------------------------- numpy parts

def degrees(N, M, HT_ptr, HT_ind):
    """hypergraph.py:34-49: degV = rowsum(H)^-1/2 with inf -> 1, shape [N,1];
    degE = colsum(H)^-1, shape [M,1], NO inf guard (float32, torch.pow)."""
    import torch  # the reference computes these with torch.pow on CPU tensors
    HT_ptr = np.asarray(HT_ptr, np.int64)
    HT_ind = np.asarray(HT_ind, np.int64)
    cntV = torch.from_numpy(np.bincount(HT_ind, minlength=N).astype(np.float64)).float()
    cntE = torch.from_numpy(np.diff(HT_ptr).astype(np.float64)).float()
    degV = cntV.pow(-0.5)
    degE = cntE.pow(-1)
    degV[torch.isinf(degV)] = 1
    return degE.reshape(M, 1).numpy(), degV.reshape(N, 1).numpy()


def read_mtx(path):
    """dataloader.hpp:22-104 semantics: banner, size line, 1-based (row, col
    [, value dropped for real/integer]) triples; `symmetric` mirrors off-diagonal
    entries then sorts and de-duplicates; `general` sorts and KEEPS duplicates.
    Returns (nrow, ncol, H_ptr, H_ind)."""
    with open(path, "r") as f:
        banner = f.readline().strip().split()
        if len(banner) < 5 or banner[0] != "%%MatrixMarket":
            raise ValueError("Could not process this file.")
        field = banner[3].lower()
        symm = banner[4].lower()
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        nrow, ncol, nnz = (int(t) for t in line.split()[:3])
        toks = f.read().split()
    per = 3 if field in ("real", "integer") else 2
    if len(toks) < per * nnz:
        raise ValueError("Error: not enough rows in mtx file.")
    arr = np.array(toks[:per * nnz]).reshape(nnz, per)
    r = arr[:, 0].astype(np.int64) - 1
    c = arr[:, 1].astype(np.int64) - 1
    if symm == "symmetric":
        off = r != c
        r, c = np.concatenate([r, c[off]]), np.concatenate([c, r[off]])
        key = np.unique(r * ncol + c)
    else:
        key = np.sort(r * ncol + c, kind="stable")
    r, c = key // ncol, key % ncol
    ptr = np.zeros(nrow + 1, np.int64)
    np.add.at(ptr, r + 1, 1)
    ptr = np.cumsum(ptr).astype(np.int32)
    return nrow, ncol, ptr, c.astype(np.int32)
