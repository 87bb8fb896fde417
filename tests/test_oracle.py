"""CPU-only: the oracle against the reference's golden vectors and against
independent formulations.  (-m "not gpu")"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, golden_files, vertex_csr
from hypergef_amd import synth


@pytest.mark.parametrize("fname", golden_files("balancer_"))
def test_balance_schedule_matches_reference_golden(oracle, fname):
    """Fixtures were produced by the reference's HyperGsys/balancer.py."""
    g = np.load(os.path.join(GOLDEN, fname))
    key, row, st, ed = oracle.balance_schedule(int(g["ngs"]), g["csrptr"])
    np.testing.assert_array_equal(key, g["balan_key"])
    np.testing.assert_array_equal(row, g["balan_row"])
    np.testing.assert_array_equal(st, g["group_st"])
    np.testing.assert_array_equal(ed, g["group_ed"])


def test_balance_schedule_toy_values(oracle):
    # csrptr=[0,3,3,8,9], ngs=2
    key, row, st, ed = oracle.balance_schedule(2, [0, 3, 3, 8, 9])
    assert key.tolist() == [0, 2, 3, 5, 7, 8, 9]
    assert row.tolist() == [0, 0, 0, 0] + [2] * 9 + [3]
    assert st.tolist() == [0, 1, 0, 1, 2, 3, 4, 2, 3, 4, 2, 3, 4, 5]
    assert ed.tolist() == [0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5]


def test_balance_schedule_empty_raises(oracle):
    with pytest.raises(IndexError):
        oracle.balance_schedule(4, [0, 0, 0])


def _scipy_H(inc):
    e = np.repeat(np.arange(inc.M), np.diff(inc.csrptr))
    return sp.csr_matrix((np.ones(inc.nnz, np.float64), (inc.colind, e)), shape=(inc.N, inc.M))


def test_hand_case(oracle):
    # 4 vertices, 3 hyperedges: e0={0,1,2}, e1={} (empty), e2={2,3}; vertex ids as features
    inc = synth.Incidence(4, 3, [0, 3, 3, 5], [0, 1, 2, 2, 3])
    X = np.array([[1, 10], [2, 20], [4, 40], [8, 80]], np.float32)
    H_ptr, H_ind = vertex_csr(inc, oracle)
    assert H_ptr.tolist() == [0, 1, 2, 4, 5] and H_ind.tolist() == [0, 0, 0, 2, 2]
    Y = oracle.hyperaggr_host(4, 2, H_ptr, H_ind, inc.csrptr, inc.colind, X)
    # Xe = [7,70], [0,0], [12,120]
    np.testing.assert_array_equal(Y, np.array([[7, 70], [7, 70], [19, 190], [12, 120]], np.float32))
    degE, degV = oracle.degrees(4, 3, inc.csrptr, inc.colind)
    np.testing.assert_allclose(degE.ravel(), [1 / 3, np.inf, 0.5])
    np.testing.assert_allclose(degV.ravel(), [1, 1, 2 ** -0.5, 1])
    W = np.array([2, 3, 5], np.float32)
    Yw = oracle.hgnn_check(4, 3, 2, H_ptr, H_ind, inc.csrptr, inc.colind, X, degE, degV, W)
    xe0, xe2 = np.float32(7) * np.float32(1 / 3) * 2, np.float32(12) * np.float32(0.5) * 5
    np.testing.assert_allclose(Yw[:, 0], [xe0, xe0, (xe0 + xe2) * np.float32(2 ** -0.5), xe2], rtol=1e-6)


def test_isolated_vertex_degree_guard(oracle):
    inc = synth.Incidence(3, 1, [0, 2], [0, 2])  # vertex 1 isolated
    degE, degV = oracle.degrees(3, 1, inc.csrptr, inc.colind)
    assert degV[1, 0] == 1.0 and degE[0, 0] == 0.5


@pytest.mark.parametrize("F", [2, 32, 128])
def test_fused_and_twostep_host_agree_bitwise(oracle, F):
    """the two reference CPU paths agree bit for bit."""
    inc = synth.cora_shape()
    H_ptr, H_ind = vertex_csr(inc, oracle)
    X = synth.features_like_reference(inc.N, F, seed=5)
    Yf = oracle.hyperaggr_host(inc.N, F, H_ptr, H_ind, inc.csrptr, inc.colind, X)
    Yt, _ = oracle.twostep_host(inc.N, inc.M, F, H_ptr, H_ind, inc.csrptr, inc.colind, X)
    assert np.array_equal(Yf, Yt)
    Yo = oracle.hyperaggr_host(inc.N, F, H_ptr, H_ind, inc.csrptr, inc.colind, X, omp=True)
    assert np.array_equal(Yf, Yo)
    # unweighted hgnn_check is the same arithmetic
    Yc = oracle.hgnn_check(inc.N, inc.M, F, H_ptr, H_ind, inc.csrptr, inc.colind, X)
    assert np.array_equal(Yf, Yc)


@pytest.mark.parametrize("shape", ["cora", "citeseer", "pubmed", "ragged"])
def test_against_scipy_float64(oracle, shape):
    """Independent formulation: Dv H De W H^T X in float64 via scipy."""
    inc = {"cora": synth.cora_shape, "citeseer": synth.citeseer_shape, "pubmed": synth.pubmed_shape,
           "ragged": lambda: synth.random_incidence(500, 300, 6.0, seed=3, empty_frac=0.1)}[shape]()
    F = 8
    rng = np.random.default_rng(11)
    X = rng.standard_normal((inc.N, F)).astype(np.float32)
    W = rng.random(inc.M).astype(np.float32) + 0.5
    H_ptr, H_ind = vertex_csr(inc, oracle)
    degE, degV = oracle.degrees(inc.N, inc.M, inc.csrptr, inc.colind)
    Y = oracle.hgnn_check(inc.N, inc.M, F, H_ptr, H_ind, inc.csrptr, inc.colind, X, degE, degV, W)
    H = _scipy_H(inc)
    dE = np.where(np.isinf(degE.ravel()), 0.0, degE.ravel().astype(np.float64))
    ref = sp.diags(degV.ravel().astype(np.float64)) @ (H @ (sp.diags(dE * W) @ (H.T @ X.astype(np.float64))))
    assert np.all(np.isfinite(Y))
    np.testing.assert_allclose(Y, ref, rtol=2e-5, atol=2e-6)
    # the device-kernel arithmetic order differs only by roundings
    Yk = oracle.hgnn_kernel_order(inc.N, inc.M, F, inc.csrptr, inc.colind, X, degE, degV, W)
    np.testing.assert_allclose(Yk, ref, rtol=2e-5, atol=2e-6)


def test_mean_max_against_numpy(oracle):
    inc = synth.random_incidence(200, 120, 5.0, seed=9)
    inc = synth.Incidence(inc.N, inc.M, inc.csrptr, inc.colind)
    F = 3
    rng = np.random.default_rng(2)
    X = rng.standard_normal((inc.N, F)).astype(np.float32)
    degE, degV = oracle.degrees(inc.N, inc.M, inc.csrptr, inc.colind)
    W = np.ones(inc.M, np.float32)
    Ym = oracle.hgnn_mean(inc.N, inc.M, F, inc.csrptr, inc.colind, X, degE, degV, W)
    Yx, rec = oracle.hgnn_max(inc.N, inc.M, F, inc.csrptr, inc.colind, X, degE, degV, W)
    ref_m = np.zeros((inc.N, F)); ref_x = np.zeros((inc.N, F))
    for e in range(inc.M):
        mem = inc.colind[inc.csrptr[e]:inc.csrptr[e + 1]]
        if len(mem) == 0:
            continue
        xe = X[mem].astype(np.float64)
        ref_m[mem] += (xe.sum(0) * degE[e, 0] / len(mem))[None, :] * degV[mem]
        ref_x[mem] += (xe.max(0) * degE[e, 0])[None, :] * degV[mem]
        assert np.array_equal(rec[e], mem[xe.argmax(0)])
    np.testing.assert_allclose(Ym, ref_m, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(Yx, ref_x, rtol=1e-5, atol=1e-6)


def test_read_mtx_semantics(oracle, tmp_path):
    """dataloader.hpp:22-104: 1-based, values dropped, sorted, symmetric mirrored+deduped."""
    p = tmp_path / "g.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n% c\n3 4 5\n"
                 "3 1 1.0\n1 2 1.0\n1 1 1.0\n2 4 1.0\n1 2 1.0\n")
    nrow, ncol, ptr, ind = oracle.read_mtx(str(p))
    assert (nrow, ncol) == (3, 4)
    assert ptr.tolist() == [0, 3, 4, 5] and ind.tolist() == [0, 1, 1, 3, 0]  # duplicate kept
    q = tmp_path / "s.mtx"
    q.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n2 1\n3 3\n2 1\n")
    nrow, ncol, ptr, ind = oracle.read_mtx(str(q))
    assert ptr.tolist() == [0, 1, 2, 3] and ind.tolist() == [1, 0, 2]  # mirrored, deduped
    # round trip of the generator's writer
    inc = synth.citeseer_shape()
    r = tmp_path / "c.mtx"
    synth.write_mtx(str(r), inc)
    nrow, ncol, H_ptr, H_ind = oracle.read_mtx(str(r))
    HT_ptr, HT_ind = oracle.transpose_csr(nrow, ncol, H_ptr, H_ind)
    assert np.array_equal(HT_ptr, inc.csrptr) and np.array_equal(HT_ind, inc.colind)


def test_check_result_is_relative_only(oracle):
    ref = np.array([[1.0, 0.0]], np.float32)
    assert oracle.check_result(np.array([[1.005, 0.0]], np.float32), ref)
    assert not oracle.check_result(np.array([[1.02, 0.0]], np.float32), ref)
    assert not oracle.check_result(np.array([[1.0, 1e-9]], np.float32), ref)  # ref == 0 demands exact 0


@pytest.mark.parametrize("shape", ["cora", "pubmed", "ragged"])
def test_against_the_scatter_formulation_in_torch(oracle, shape):
    """A third independent statement of the operator: the reference's PyG layer (model/pygnn/hgnn.py:25-38) --
    gather X[vertex], scatter-sum by hyperedge, scale by degE then W, gather Xe[edges], scatter-sum by vertex,
    scale by degV -- written with torch index_add_ on the CPU (torch_scatter's sum is the same accumulation).  Same
    fp32 arithmetic in the same order as the oracle's HGNN_check restatement: it must agree to the last bit or two."""
    import torch
    inc = {"cora": synth.cora_shape, "pubmed": synth.pubmed_shape,
           "ragged": lambda: synth.random_incidence(500, 300, 6.0, seed=3, empty_frac=0.1)}[shape]()
    F = 6
    rng = np.random.default_rng(12)
    X = rng.standard_normal((inc.N, F)).astype(np.float32)
    W = (rng.random(inc.M) + 0.5).astype(np.float32)
    H_ptr, H_ind = vertex_csr(inc, oracle)
    degE, degV = oracle.degrees(inc.N, inc.M, inc.csrptr, inc.colind)
    degE = np.where(np.isinf(degE), 0, degE).astype(np.float32)
    vertex = torch.from_numpy(inc.colind.astype(np.int64))
    edges = torch.from_numpy(np.repeat(np.arange(inc.M, dtype=np.int64), np.diff(inc.csrptr)))
    Xt = torch.from_numpy(X)
    Xe = torch.zeros(inc.M, F).index_add_(0, edges, Xt[vertex])
    Xe = Xe * torch.from_numpy(degE.reshape(-1, 1))
    Xe = Xe * torch.from_numpy(W.reshape(-1, 1))
    Xv = torch.zeros(inc.N, F).index_add_(0, vertex, Xe[edges]) * torch.from_numpy(degV.reshape(-1, 1))
    Y = oracle.hgnn_check(inc.N, inc.M, F, H_ptr, H_ind, inc.csrptr, inc.colind, X, degE, degV, W)
    np.testing.assert_allclose(Y, Xv.numpy(), rtol=3e-7, atol=1e-7)
