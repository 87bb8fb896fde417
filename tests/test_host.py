"""CPU-only: the C-ABI library loads, exports every declared symbol, and its
host logic (reference-compatible scheduler, plan builder) is right.  No device
compute is called here.  (-m "not gpu")"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_files, vertex_csr
from hypergef_amd import synth


def test_library_exports_every_declared_symbol(hg):
    from hypergef_amd import _lib
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "hg_aggr.h")).read()
    declared = set(re.findall(r"HG_API[^;(]*?\b(hg_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.hg_version() == 410  # HG_AGGR_VERSION: round 4 added hg_aggr_linear_res_dev_f32, then HG_LIN_BF16X6 + hg_linear_pack_ex_f32
    assert L.hg_status_string(-4) == b"workspace too small"


@pytest.mark.parametrize("fname", golden_files("balancer_"))
def test_native_balance_schedule_matches_reference_golden(hg, fname):
    g = np.load(os.path.join(GOLDEN, fname))
    bs = hg.balance_schedule(int(g["ngs"]), g["csrptr"])
    np.testing.assert_array_equal(bs.balan_key, g["balan_key"])
    np.testing.assert_array_equal(bs.balan_row, g["balan_row"])
    np.testing.assert_array_equal(bs.group_st, g["group_st"])
    np.testing.assert_array_equal(bs.group_ed, g["group_ed"])
    assert bs.balan_key.dtype == np.int32 and bs.nrow == g["csrptr"].shape[0] - 1


def test_native_balance_schedule_accepts_torch_and_rejects_empty(hg):
    import torch
    bs = hg.balance_schedule(2, torch.tensor([0, 3, 3, 8, 9], dtype=torch.int32))
    assert bs.balan_key.tolist() == [0, 2, 3, 5, 7, 8, 9]
    with pytest.raises(IndexError):
        hg.balance_schedule(4, np.array([0, 0, 0], np.int32))


def test_balance_schedule_no_float32_rounding(hg):
    """Defect D7: indices above 2**24 must survive (reference rounds them)."""
    base = (1 << 24) + 1
    csrptr = np.array([0, base, base + 3], np.int64).astype(np.int32)
    bs = hg.balance_schedule(1 << 30, csrptr)
    assert bs.balan_key.tolist() == [0, base, base + 3]


def _emulate(plan_sched, ptr, ind, src, scaleA, scaleB, F):
    """numpy float32 emulation of gather_rows_kernel + fixup semantics."""
    nrows = ptr.shape[0] - 1
    dst = np.full((nrows, F), np.nan, np.float32)
    written = np.zeros(nrows, np.int32)

    def scale(r, acc):
        if scaleA is not None:
            acc = acc * scaleA[r]
        if scaleB is not None:
            acc = acc * scaleB[r]
        return acc

    for row0, n, nnz0, cnt in plan_sched["panels"]:
        assert ptr[row0] == nnz0 and ptr[row0 + n] - nnz0 == cnt
        for r in range(row0, row0 + n):
            acc = np.zeros(F, np.float32)
            for p in range(ptr[r], ptr[r + 1]):
                acc = acc + src[ind[p]]
            dst[r] = scale(r, acc) if ptr[r + 1] > ptr[r] else acc
            written[r] += 1
    nslots = int(plan_sched["fixups"][:, 2].sum()) if len(plan_sched["fixups"]) else 0
    partial = np.full((nslots, F), np.nan, np.float32)
    for row, beg, end, slot in plan_sched["tasks"]:
        assert ptr[row] <= beg < end <= ptr[row + 1]
        acc = src[ind[beg:end]].astype(np.float64).sum(0).astype(np.float32)
        if slot < 0:
            assert beg == ptr[row] and end == ptr[row + 1]
            dst[row] = scale(row, acc)
            written[row] += 1
        else:
            partial[slot] = acc
    for row, first, count, out in plan_sched["fixups"]:  # first-level fixups come first
        acc = np.zeros(F, np.float32)
        for k in range(count):
            acc = acc + partial[first + k]
        if out > 0:  # two-level sum: this run of slots goes to a slot of its own
            partial[out - 1] = acc
            continue
        dst[row] = scale(row, acc)
        written[row] += 1
    assert np.all(written == 1), "every row must be produced exactly once"
    return dst


def _hub_incidence(N=300, M=260, seed=5):
    """Vertex 0 belongs to every hyperedge: its row of H is cut into far more tasks than one
    fixup sums, so the schedule must use the two-level form."""
    rng = np.random.default_rng(seed)
    rows = [np.unique(np.concatenate([[0], rng.integers(1, N, rng.integers(1, 6))])) for _ in range(M)]
    ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    return synth.Incidence(N, M, ptr, np.concatenate(rows).astype(np.int32), name="hub")


@pytest.mark.parametrize("case", ["cora", "pubmed", "ragged_split", "tiny_panels", "two_level"])
def test_plan_schedule_covers_and_reproduces_oracle(hg, oracle, case):
    from hypergef_amd import plan as planmod
    if case == "cora":
        inc, opts = synth.cora_shape(), planmod.make_opts(host_only=True)
    elif case == "pubmed":
        inc, opts = synth.pubmed_shape(), planmod.make_opts(host_only=True)
    elif case == "ragged_split":  # long rows cut into several tasks + fixups, empty hyperedges
        inc = synth.random_incidence(400, 150, 14.0, seed=4, empty_frac=0.15)
        opts = planmod.make_opts(short_max=6, split_len=8, panel_rows=16, panel_nnz=32, host_only=True)
    elif case == "two_level":
        inc = _hub_incidence()
        opts = planmod.make_opts(short_max=4, split_len=4, panel_rows=16, panel_nnz=32, host_only=True)
    else:
        inc = synth.random_incidence(97, 211, 2.5, seed=8, empty_frac=0.3)
        opts = planmod.make_opts(short_max=4, split_len=4, panel_rows=3, panel_nnz=7, host_only=True)
    plan = planmod.Plan.from_host(inc.N, inc.M, inc.csrptr, inc.colind, opts)
    H_ptr, H_ind = vertex_csr(inc, oracle)
    pv, iv = plan.vertex_csr()
    assert np.array_equal(pv, H_ptr) and np.array_equal(iv, H_ind)

    F = 3
    rng = np.random.default_rng(1)
    X = rng.standard_normal((inc.N, F)).astype(np.float32)
    degE, degV = oracle.degrees(inc.N, inc.M, inc.csrptr, inc.colind)
    W = (rng.random(inc.M) + 0.5).astype(np.float32)
    s0, s1 = plan.schedule(0), plan.schedule(1)
    info = plan.info
    for hop, s in ((0, s0), (1, s1)):
        lens = np.diff(inc.csrptr if hop == 0 else H_ptr)
        assert info["max_len"][hop] == (lens.max() if len(lens) else 0)
        if len(s["panels"]):
            assert s["panels"][:, 1].max() <= info["panel_rows"]
            assert s["panels"][:, 3].max() <= info["panel_nnz"]
        if len(s["tasks"]):
            tl = s["tasks"][:, 2] - s["tasks"][:, 1]
            assert tl.max() <= info["split_len"] and np.all(np.diff(tl) <= 0)  # longest first
        assert info["partials"][hop] == (s["fixups"][:, 2].sum() if len(s["fixups"]) else 0)
        if len(s["fixups"]):
            lvl1 = s["fixups"][:, 3] > 0
            assert s["fixups"][:, 2].max() <= 32  # no serial chain longer than the fan-in
            assert not np.any(np.diff(lvl1.astype(np.int8)) > 0)  # first-level fixups come first
            assert (case == "two_level" and hop == 1) == bool(lvl1.any())
    Xe = _emulate(s0, inc.csrptr, inc.colind, X, degE.ravel(), W, F)
    Y = _emulate(s1, H_ptr, H_ind, Xe, degV.ravel(), None, F)
    Yref, Xe_ref = oracle.hgnn_check(inc.N, inc.M, F, H_ptr, H_ind, inc.csrptr, inc.colind, X,
                                     degE, degV, W, return_xe=True)
    live = np.diff(inc.csrptr) > 0
    np.testing.assert_allclose(Xe[live], Xe_ref[live], rtol=1e-5, atol=1e-6)
    assert np.all(Xe[~live] == 0)  # empty hyperedges stay exactly 0 even though degE is inf
    np.testing.assert_allclose(Y, Yref, rtol=1e-5, atol=1e-6)
    short = np.diff(inc.csrptr) <= info["short_max"]
    assert np.array_equal(Xe[short & live], Xe_ref[short & live])  # panel rows are bit-exact


def test_plan_rejects_bad_input(hg):
    from hypergef_amd import _lib, plan as planmod
    ho = planmod.make_opts(host_only=True)
    with pytest.raises(_lib.HgError):  # column index out of range
        planmod.Plan.from_host(3, 2, np.array([0, 1, 2], np.int32), np.array([0, 3], np.int32), ho)
    with pytest.raises(_lib.HgError):  # non-monotone row pointer
        planmod.Plan.from_host(3, 2, np.array([0, 2, 1], np.int32), np.array([0], np.int32), ho)
    with pytest.raises(_lib.HgError):  # short_max > panel_nnz
        planmod.Plan.from_host(3, 1, np.array([0, 1], np.int32), np.array([0], np.int32),
                               planmod.make_opts(short_max=64, panel_nnz=32, host_only=True))
    with pytest.raises(ValueError):
        planmod.Plan.from_host(3, 2, np.array([0, 1], np.int32), np.array([0], np.int32), ho)


def test_host_only_plan_refuses_to_launch(hg):
    from hypergef_amd import _lib, plan as planmod
    inc = synth.cora_shape()
    plan = planmod.Plan.from_host(inc.N, inc.M, inc.csrptr, inc.colind, planmod.make_opts(host_only=True))
    ws = plan.workspace_bytes(32)
    assert ws >= inc.M * 32 * 4 and ws % 256 == 0
    rc = _lib.lib().hg_aggr_fused_f32(plan._h, 32, None, None, None, None, None, None, None, None, 0, 0, None)
    assert rc == -1  # HG_ERR_INVALID, never a crash


def test_degenerate_plans(hg):
    from hypergef_amd import plan as planmod
    ho = planmod.make_opts(host_only=True)
    p = planmod.Plan.from_host(5, 0, np.array([0], np.int32), np.zeros(0, np.int32), ho)
    assert p.info["panels"] == [0, 1] and p.nnz == 0
    p = planmod.Plan.from_host(4, 3, np.array([0, 0, 0, 0], np.int32), np.zeros(0, np.int32), ho)
    assert p.info["panels"] == [1, 1] and p.info["tasks"] == [0, 0]


def test_ops_refuse_cpu_tensors(hg):
    """The product path never computes on the CPU."""
    import torch
    inc = synth.cora_shape()
    x = torch.zeros(inc.N, 4)
    ptr, ind = torch.from_numpy(inc.csrptr), torch.from_numpy(inc.colind)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hg.ops.unignnaggr(ptr, ptr, ptr, ptr, ptr, ind, x)


def test_native_mtx_reader_matches_oracle(hg, oracle, tmp_path):
    """hg_mtx_read vs the oracle's restatement of dataloader.hpp:22-141."""
    from hypergef_amd.mtx import read_mtx
    cases = {
        "general_dupes.mtx": "%%MatrixMarket matrix coordinate real general\n% c\n3 4 5\n"
                             "3 1 1.0\n1 2 1.0\n1 1 1.0\n2 4 1.0\n1 2 1.0\n",
        "symmetric.mtx": "%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n2 1\n3 3\n2 1\n",
        "integer.mtx": "%%MatrixMarket matrix coordinate integer general\n2 2 2\n2 2 7\n1 1 3\n",
    }
    for fname, text in cases.items():
        p = tmp_path / fname
        p.write_text(text)
        inc, (H_ptr, H_ind) = read_mtx(p)
        nrow, ncol, o_ptr, o_ind = oracle.read_mtx(str(p))
        assert (inc.N, inc.M) == (nrow, ncol)
        assert np.array_equal(H_ptr, o_ptr) and np.array_equal(H_ind, o_ind)
        t_ptr, t_ind = oracle.transpose_csr(nrow, ncol, o_ptr, o_ind)
        assert np.array_equal(inc.csrptr, t_ptr) and np.array_equal(inc.colind, t_ind)
    src = synth.pubmed_shape()
    q = tmp_path / "pubmed.mtx"
    synth.write_mtx(str(q), src)
    inc, _ = read_mtx(q)
    assert np.array_equal(inc.csrptr, src.csrptr) and np.array_equal(inc.colind, src.colind)
    from hypergef_amd import _lib
    with pytest.raises(_lib.HgError, match="not found"):
        read_mtx(tmp_path / "missing.mtx")
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n")
    with pytest.raises(_lib.HgError, match="not enough rows"):
        read_mtx(bad)


def test_aggr_proto_cli_cpu_mode(hg, tmp_path):
    """BASELINE.json configs[0]: `aggr_proto <mtx> 32` plumbing on the CPU path, no GPU."""
    import subprocess
    exe = os.path.join(ROOT, "bin", "aggr_proto")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "hypergef_amd", "csrc"), "cli"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and r.stdout.startswith("Input: first get the path of sparse matrix")
    mtx = tmp_path / "cora.mtx"
    inc = synth.cora_shape()
    synth.write_mtx(str(mtx), inc)
    r = subprocess.run([exe, str(mtx), "32", "--cpu"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == "H.nrow %d H.ncol %d H_nnz %d" % (inc.N, inc.M, inc.nnz)
    assert "bitwise equal" in lines[1]
    assert "start spmm test" in lines and "start fused kernel test" in lines and lines.count("check passed!") == 2
    row = (tmp_path / "result.csv").read_text().strip().split(",")
    assert row[0] == str(mtx) and row[1] == "32" and len(row) == 10


def test_models_torch_backend_cpu(hg):
    """The index_add_ baseline networks build and train on CPU (they stand in for PyG/DGL)."""
    import argparse
    import torch
    from hypergef_amd import models
    inc = synth.citeseer_shape()
    hyperg = hg.HyperGraph.from_incidence(inc, "cpu", data_name="citeseer")
    assert hyperg.group_key.dtype == torch.int32 and hyperg.ngs == 6
    args = argparse.Namespace(model="HGNN", activation="relu", input_drop=0.0, dropout=0.0, backend="torch",
                              device="cpu")
    torch.manual_seed(0)
    X = torch.randn(inc.N, 12)
    y = torch.randint(0, 4, (inc.N,))
    for name in ("HGNN", "UniGIN", "UniGCNII"):
        args.model = name
        net = (models.UniGCNII(args, hyperg, 12, 8, 4, 2, 1) if name == "UniGCNII"
               else models.HGsysHGNN(args, hyperg, 12, 8, 4, 2, "sum", 1))
        opt = torch.optim.Adam(net.parameters(), lr=0.01)
        losses = []
        for _ in range(15):
            opt.zero_grad()
            loss = torch.nn.functional.nll_loss(net(X), y)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_schedule_builders_under_sanitizers(tmp_path):
    """AddressSanitizer + UBSan over the host-side code (schedule builders on random hypergraphs,
    balance_schedule, the MatrixMarket reader on good and broken files)
    (tests/native/sched_fuzz.cpp); CPU build only -- the GPU pool has no sanitizer runs."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sched_fuzz")
    src = [os.path.join(root, "tests", "native", "sched_fuzz.cpp"),
           os.path.join(root, "hypergef_amd", "csrc", "hg_schedule.cpp"),
           os.path.join(root, "hypergef_amd", "csrc", "hg_fused.cpp"),
           os.path.join(root, "hypergef_amd", "csrc", "hg_mtx.cpp")]
    cc = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                         "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "hypergef_amd", "csrc"),
                         "-o", exe] + src, capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr:
        pytest.skip("g++ has no sanitizer runtime here")
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe, str(tmp_path / "fuzz.mtx")], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "sched_fuzz ok" in run.stdout, (run.stdout + run.stderr)[-3000:]


def test_auto_variant_decision_is_cached(hg):
    """HG_VARIANT_AUTO is resolved on every aggregation call: after the first call for a feature
    width it must cost nothing (it once re-walked the whole graph per call: 1 ms on the bench batch)."""
    import time
    from hypergef_amd import plan as planmod
    inc = synth.replicate_block_diagonal(synth.cora_shape(), 128)
    plan = planmod.Plan.from_host(inc.N, inc.M, inc.csrptr, inc.colind, planmod.make_opts(host_only=True))
    first = plan.auto_variant(32)
    t0 = time.perf_counter()
    for _ in range(200):
        assert plan.auto_variant(32) == first
    per_call = (time.perf_counter() - t0) / 200
    assert per_call < 50e-6, "auto variant costs %.1f us per call" % (per_call * 1e6)


def test_launch_bound_graphs_get_small_panels_and_sub_slots(hg):
    """One dataset-sized hypergraph is launch-bound: the plan sizes its panels for that (more, shorter
    workgroups) and cuts long hyperedges into sub-slots instead of materialising them (one launch);
    a batch in the throughput regime, or a caller who fixes the tile, keeps the 16 KB tile."""
    from hypergef_amd import plan as planmod
    host = planmod.make_opts(host_only=True)
    cora = synth.cora_shape()
    one = planmod.Plan.from_host(cora.N, cora.M, cora.csrptr, cora.colind, host).prepare(32)
    assert one["cap"] < 128 and one["panels"] > 50 and one["n_mat"] == 0 and one["fixups"] == 0
    fixed = planmod.Plan.from_host(cora.N, cora.M, cora.csrptr, cora.colind,
                                   planmod.make_opts(host_only=True, fused_tile_bytes=16384)).prepare(32)
    assert fixed["cap"] == 128
    batch = synth.replicate_block_diagonal(cora, 64)
    assert planmod.Plan.from_host(batch.N, batch.M, batch.csrptr, batch.colind, host).prepare(32)["cap"] == 128
    cs = synth.citeseer_shape()  # hyperedges of up to 26 members
    p = planmod.Plan.from_host(cs.N, cs.M, cs.csrptr, cs.colind, host)
    info = p.prepare(32)
    assert info["n_mat"] == 0 and info["fixups"] == 0 and p.auto_variant(32) == "fused"
    assert info["slots"] > planmod.Plan.from_host(cs.N, cs.M, cs.csrptr, cs.colind,
                                                  planmod.make_opts(host_only=True, fused_tile_bytes=16384)).prepare(32)["slots"] // 2


def test_options_resolution_needs_no_gpu(hg):
    """ops.Options: validation, per-thread nesting of `with ops.options(...)`, process defaults through the set_*
    functions -- the resolution order every operator call uses (explicit > thread-local > defaults)."""
    import threading
    from hypergef_amd import ops
    base = ops.current_options()
    assert base == ops.Options() and base.variant == "auto" and base.backward == "reference"
    with pytest.raises(ValueError):
        ops.Options(fuse_linear="sometimes")
    with pytest.raises(ValueError):
        ops.set_backward("exact")
    with ops.options(variant="pull") as outer:
        assert ops.current_options() is outer
        with ops.options(backward="adjoint") as inner:
            assert inner.variant == "pull" and inner.backward == "adjoint" and ops.current_options() is inner
            seen = []
            t = threading.Thread(target=lambda: seen.append(ops.current_options()))
            t.start()
            t.join()
            assert seen == [base]
        assert ops.current_options() is outer
    assert ops.current_options() == base
    ops.set_fuse_linear("never")
    try:
        assert ops.current_options().fuse_linear == "never" and ops._opt(None).fuse_linear == "never"
        explicit = ops.Options(fuse_linear="always")
        assert ops._opt(explicit) is explicit
    finally:
        ops.set_fuse_linear("auto")
    with pytest.raises(TypeError):
        ops._opt("pull")
    # the linear epilogue's arithmetic: fp32 MFMA unless asked otherwise
    assert base.linear_math == "f32" and ops.Options(linear_math="bf16x6").linear_math == "bf16x6"
    with pytest.raises(ValueError):
        ops.Options(linear_math="bf16")
    ops.set_linear_math("bf16x6")
    try:
        assert ops.current_options().linear_math == "bf16x6"
        with ops.options(linear_math="f32") as o:
            assert o.linear_math == "f32"
    finally:
        ops.set_linear_math("f32")
