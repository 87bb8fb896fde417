#!/usr/bin/env python3
"""bench.py -- aggregated edges/s of the fused V->E->V aggregation on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one fused aggregation Y = H H^T X (the `aggr_proto` operator,
HyperGsys/include/hgnnAgg.cuh) over one batch of synthetic input resident in
HBM.  Workload (BASELINE.json configs[1], in the only form that leaves the
256 MiB Infinity Cache -- SURVEY.md 8(d) "C2xK"): a batch of K = 1024
cora-shape hypergraphs (N=2708, M=1579, nnz=4859 each) as one block-diagonal
incidence matrix, feat_len = 32, fp32.  The single-hypergraph latency (the
number the reference's result.xlsx reports) is printed in `single_graph`.

The timed output is checked: after the timed region Y is compared with the CPU
oracle on the same input (`parity`), and a mismatch makes the run fail -- the
reference never times an unchecked variant either (TRY, hgnnAgg.cuh:1159-1169).

`configs` (N = 1): measured the same way in the same run -- the other five cells
of north_star's target matrix (cora / citeseer / pubmed shape x F = 32 / 128;
pubmed-shape x64 at F = 128 is BASELINE config 3), config 3's MFMA path (the
aggregation with the layer's 128 -> 128 linear folded in: hg_aggr_linear_f32,
priced against the HBM and the fp32-MFMA roofline), the power-law |V|=1M |E|=4M
hypergraph at F = 64 (config 4), and the weighted operator (degE, degV, W:
HGNNConv itself) on the headline batch.  Every entry's output is checked against
the oracle and against a float64 answer (scipy) at 1e-5 * max(1, |ref|).

N > 1: one process per GPU (torchrun).  Headline: the batch sharded by hyperedge
group = by hypergraph, K graphs per rank (weak scaling); no vertex is shared
between shards, so the data path has no collective and ranks meet in the barrier
that brackets the timed region.  `strong_scaling` (beside it on the line; `sharded` in the detail file): ONE hypergraph (config 4)
cut into hyperedge groups across the ranks by hypergef_amd.dist.ShardedAggregator
-- X replicated, each rank aggregates its hyperedges into a dense partial, one
RCCL all-reduce (or reduce-scatter) of N*F*4 bytes over xGMI sums them: the
configuration north_star names, strong scaling.

Output (rank 0).  The LAST stdout line is one JSON object of at most 4 KB (`compact_line`): the contract keys,
`roofline`, `cpu_baseline`, a top-level `parity`, one short record per configuration under `configs`, the
single-hypergraph latencies beside the rocSPARSE two-step comparator, and -- N > 1 -- `strong_scaling`.  Everything
else (schedules, floors, notes, per-configuration rooflines and parity reports) goes to `bench_detail.json` next to
this file (and to gpurun_out/ when that directory exists), never to stdout: the reference's own measurement output is
one short CSV row (HyperGsys/source/aggr_proto.cu:51,82).

`python bench.py --gpus N` with N > 1 outside torchrun starts `python -m torch.distributed.run --nproc-per-node N`
on this same file as a child process (before anything touches the GPU) and exits with its code.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
U32 = 2.0 ** -24       # unit roundoff of fp32
LINE_LIMIT = 4096      # bytes of the final stdout line (the driver keeps an 8 KB tail of stdout)
# BASELINE.md 1a (result.xlsx "fig7,fig9", RTX 3090, F = 32): cuSPARSE 2xSpMM ms, best fused kernel ms
REFERENCE_RTX3090_F32 = {"cora": (0.040672, 0.0047949), "citeseer": (0.040387, 0.0036982), "pubmed": (0.057672, 0.012484)}


def b_alg(N, M, nnz, F, n_w, has_degV):
    """Algorithmic (compulsory) bytes of one fused aggregation, SURVEY.md 8(d): X read and Y
    written once, the incidence indices once per hop, both row-pointer arrays, the per-hyperedge
    scale vectors (n_w of them) and degV -- each only when the operator reads it."""
    return 4 * (2 * N * F + 2 * nnz + (M + 1) + (N + 1) + n_w * M + (N if has_degV else 0))


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--shape", default="cora", choices=["cora", "citeseer", "pubmed", "powerlaw"])
    p.add_argument("--replicas", type=int, default=1024, help="hypergraphs per GPU in the batch")
    p.add_argument("--feat", type=int, default=32)
    p.add_argument("--variant", default="auto", choices=["auto", "pull", "fused", "push_atomic"])
    p.add_argument("--t-big", type=int, default=0)
    p.add_argument("--tile-bytes", type=int, default=0)
    p.add_argument("--fused-steps", type=int, default=0)
    p.add_argument("--no-hub-pass", action="store_true")
    p.add_argument("--no-row-stream", action="store_true")
    p.add_argument("--weighted", action="store_true", help="hgnnaggr (degE, degV, W) instead of H H^T X")
    p.add_argument("--linear-out", type=int, default=0,
                   help="fold the layer's linear feat -> N into the aggregation (hg_aggr_linear_f32: the MFMA path)")
    p.add_argument("--linear-math", default="f32", choices=["f32", "bf16x6"],
                   help="with --linear-out at feat 128: the matrix phase on fp32 MFMA, or six bf16 products per fp32 product")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of the timed output")
    p.add_argument("--no-extras", action="store_true", help="skip device-copy / single-graph / sharded extras")
    p.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs (N = 1)")
    p.add_argument("--config-steps", type=int, default=100)
    p.add_argument("--short-max", type=int, default=0)
    p.add_argument("--panel-rows", type=int, default=0)
    p.add_argument("--panel-nnz", type=int, default=0)
    p.add_argument("--no-xcd-remap", action="store_true")
    p.add_argument("--sharded-nodes", type=int, default=1_000_000, help="N > 1: the one hypergraph that is sharded")
    p.add_argument("--sharded-edges", type=int, default=4_000_000)
    p.add_argument("--sharded-feat", type=int, default=64)
    p.add_argument("--sharded-chunks", type=int, default=2, help="column slices of the pipelined all-reduce")
    p.add_argument("--force-collective", action="store_true",
                   help="world size 1 under torchrun: initialise RCCL anyway and run the `sharded` section through it")
    p.add_argument("--share-gpu", action="store_true",
                   help="rehearse the N>1 path on one GPU: all ranks use cuda:0, gloo instead of RCCL")
    p.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                   help="where the full result goes (the final stdout line is the <= 4 KB summary of it)")
    p.add_argument("--no-comparator", action="store_true",
                   help="skip the rocSPARSE two-step comparator (bin/aggr_proto on single cora / citeseer / pubmed shapes)")
    p.add_argument("--rehearse-cpu", action="store_true",
                   help="control flow only, no GPU and no kernels: launch, rendezvous over gloo, max over ranks, "
                        "the output line (tests/test_bench_line.py)")
    p.add_argument("--inject-exchange-error", action="store_true", help=argparse.SUPPRESS)
    return p.parse_args(argv)


def make_workload(shape, replicas):
    from hypergef_amd import synth
    if shape == "powerlaw":
        base = synth.powerlaw(1_000_000, 4_000_000, seed=3)
        return base, base
    base = {"cora": synth.cora_shape, "citeseer": synth.citeseer_shape, "pubmed": synth.pubmed_shape}[shape]()
    return base, synth.replicate_block_diagonal(base, replicas)


def workload_name(shape, replicas, F):
    if shape == "powerlaw":
        return "power-law |V|=1M |E|=4M, F=%d" % F
    return "%s-shape x%d block-diagonal batch, F=%d" % (shape, replicas, F)


def timed_steps(fn, steps, sync, barrier):
    """K calls of fn between two barrier + synchronize pairs: wall seconds, and the device seconds
    between two HIP events recorded on the stream the kernels are launched on (torch's current
    stream is the one the library receives)."""
    import torch
    barrier()
    sync()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    sync()
    t1 = time.perf_counter()
    barrier()
    return t1 - t0, ev0.elapsed_time(ev1) * 1e-3


def oracle_pass(base, inc, F, X_host, weighted, scales, time_it):
    """The oracle on this box's host cores (rank 0, N = 1): the reference CPU path restated
    (oracle/hg_oracle.c).  Returns (reference rows or None, rows checked, cpu_baseline dict or None).
    Unweighted batches: util::hyperaggr_reference_host (check.cuh:83-114) over the whole batch, timed
    for about 10 s.  One big hypergraph: TwostepSpMM_host (spmm.cuh:724-740), one pass (the fused host
    path costs sum_v sum_e |e| row reads there: minutes).  Weighted: HGNN_check's arithmetic
    (test/hgnn_test.py:56-63) on a prefix of the batch, checker only."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    blocks = inc.M // base.M
    if weighted:
        S = min(blocks, 64)
        Ms, Ns = base.M * S, base.N * S
        ptr = inc.csrptr[:Ms + 1]
        ind = inc.colind[:ptr[-1]]
        H_ptr, H_ind = orc.transpose_csr(Ms, Ns, ptr, ind)
        degE, degV, W = scales
        ref = orc.hgnn_check(Ns, Ms, F, H_ptr, H_ind, ptr, ind, np.ascontiguousarray(X_host[:Ns]),
                             degE[:Ms], degV[:Ns], W[:Ms])
        return ref, Ns, None
    if blocks == 1 and inc.nnz > 2_000_000:
        H_ptr, H_ind = orc.transpose_csr(inc.M, inc.N, inc.csrptr, inc.colind)
        t0 = time.perf_counter()
        ref, _ = orc.twostep_host(inc.N, inc.M, F, H_ptr, H_ind, inc.csrptr, inc.colind, X_host)
        dt = time.perf_counter() - t0
        cpu = {"value": inc.nnz / dt, "unit": "edges/s", "cores": 1, "kind": "port",
               "sample": "whole %s hypergraph, F=%d, one pass of the two-step host path (%.2f s)"
                         % (base.name, F, dt)}
        return ref, inc.N, cpu
    S = min(blocks, 1024)
    Ms, Ns = base.M * S, base.N * S
    ptr = inc.csrptr[:Ms + 1]
    ind = inc.colind[:ptr[-1]]
    H_ptr, H_ind = orc.transpose_csr(Ms, Ns, ptr, ind)
    Xs = np.ascontiguousarray(X_host[:Ns])
    nnz_s = int(ptr[-1])
    best, passes, ref = None, 0, None
    t_all = time.perf_counter()
    budget = 10 if time_it else 0
    while passes < (3 if time_it else 1) or (time.perf_counter() - t_all < budget and passes < 50):
        t0 = time.perf_counter()
        ref = orc.hyperaggr_host(Ns, F, H_ptr, H_ind, ptr, ind, Xs)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        passes += 1
        if time.perf_counter() - t_all > 30:
            break
    if not time_it:
        return ref, Ns, None
    cpu = {"value": nnz_s / best, "unit": "edges/s", "cores": 1, "kind": "port",
           "sample": "first %d of %d %s hypergraphs of the batch, F=%d, best of %d passes (%.3f s each)"
                     % (S, blocks, base.name, F, passes, best)}
    try:
        threads = orc.num_threads()
        best_mt = None
        for _ in range(6):  # first pass warms the thread pool
            t0 = time.perf_counter()
            orc.hyperaggr_host(Ns, F, H_ptr, H_ind, ptr, ind, Xs, omp=True)
            dt = time.perf_counter() - t0
            best_mt = dt if best_mt is None else min(best_mt, dt)
        cpu["all_cores"] = {"value": nnz_s / best_mt, "cores": threads}
        cpu["host"] = "%d logical cpus" % (os.cpu_count() or 0)
    except Exception as exc:  # the baseline is informative; never fail the bench on it
        cpu["all_cores_error"] = str(exc)
    return ref, Ns, cpu


FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, exact f32, 64 FLOP/clk/SIMD


def float64_answer(inc, X_host, scales, weight=None):
    """Dv H De W H^T X in float64 (scipy CSR products): the answer both the kernels and the oracle's fp32
    chains approximate.  With `weight` ([F_out, F_in]): (that) . weight^T."""
    import numpy as np
    import scipy.sparse as sp
    HT = sp.csr_matrix((np.ones(inc.nnz, np.float64), inc.colind, inc.csrptr), shape=(inc.M, inc.N))
    Xe = HT @ X_host.astype(np.float64)
    if scales is not None:
        degE, degV, W = scales
        Xe *= degE.astype(np.float64)[:, None]
        Xe *= W.astype(np.float64)[:, None]
    Y = HT.T.tocsr() @ Xe
    if scales is not None:
        Y *= scales[1].astype(np.float64)[:, None]
    if weight is not None:
        Y = Y @ weight.astype(np.float64).T
    return Y


def float64_report(Y_dev, inc, X_host, scales, weight=None):
    """Every element of the timed output against the float64 answer.  Plain aggregation: north_star's literal
    tolerance, 1e-5 * max(1, |ref|) -- no allowance for chain lengths.  With the linear folded in (`weight`): 1e-5 *
    max(1, l1 mass of the element), the mass being |Dv| H |De W| H^T |X| |Wlin|^T -- the error of a K-term fp32
    product chain is relative to the size of its terms, not to a result that may have cancelled (Wlin is signed), and
    the bound is per element: a small output next to a large one gets no slack from it."""
    import numpy as np
    y64 = float64_answer(inc, X_host, scales, weight)
    y = Y_dev.cpu().numpy().astype(np.float64)
    if weight is None:
        den = np.maximum(1.0, np.abs(y64))
        bound = "1e-5*max(1,|ref|)"
    else:
        abs_scales = None if scales is None else tuple(np.abs(s) for s in scales)
        den = np.maximum(1.0, float64_answer(inc, np.abs(X_host), abs_scales, np.abs(weight)))
        bound = "1e-5*max(1, (|A||X||Wlin|^T)[v,k]) per element"
    err = np.abs(y - y64) / den
    return {"max_rel_err_vs_float64": float(err.max()), "float64_ok": bool(err.max() <= 1e-5),
            "float64_violations": int((err > 1e-5).sum()),
            "float64_rows_checked": int(inc.N), "float64_bound": bound}, den


def floor_of(roofline, copy_gbs):
    """What the bytes this schedule moves (PMC traffic, profiles/traffic.json) cost at the rate the guide calls
    achievable (6.3 TB/s) and at the rate a plain device copy reached in this very run: the step time below which
    this decomposition cannot go, next to the time it takes."""
    t = roofline.get("traffic")
    if not t:
        return
    balg = roofline["algorithmic_bytes_per_step"]
    roofline["floor"] = {
        "traffic_over_algorithmic": t / balg,
        "ms_at_6300_GBs": t / 6.3e12 * 1e3, "frac_at_6300_GBs": balg / (t / 6.3e12) / 1e9 / HBM_PEAK_GBS,
        "device_copy_GBs_this_run": copy_gbs,
        "ms_at_device_copy_rate": t / (copy_gbs * 1e9) * 1e3,
        "frac_at_device_copy_rate": balg / (t / (copy_gbs * 1e9)) / 1e9 / HBM_PEAK_GBS,
        "note": "PMC bytes of this schedule / a bandwidth: the time (and fraction of the 8 TB/s roofline on algorithmic "
                "bytes) this decomposition cannot beat; measured avg_step_us is beside it"}


def parity_report(Y_dev, ref, nrows, inc):
    """Timed output vs the oracle.  Bound per row: 1e-5 * max(1, |ref|) (BASELINE.json north_star)
    plus the oracle's own worst-case rounding, u * (longest sequential chain feeding the row) --
    nothing for the short rows of the dataset shapes, and what makes a vertex in 10^6 hyperedges
    comparable at all: there the oracle's single fp32 chain is the less accurate side (it is up to
    1e-3 off the float64 answer; the kernels' blocked sums stay within 1e-5 of it,
    tests/test_gpu_parity.py::test_config4_powerlaw_full_size)."""
    import numpy as np
    y = Y_dev[:nrows].cpu().numpy()
    sizes = np.diff(inc.csrptr).astype(np.int64)
    deg = np.bincount(inc.colind, minlength=inc.N).astype(np.int64)[:nrows]
    chain = deg + int(sizes.max() if sizes.size else 0)
    err = np.abs(y - ref) / np.maximum(1.0, np.abs(ref))
    tol = 1e-5 + U32 * np.where(chain > 64, chain, 0)[:, None]
    bad = err > tol
    return {"ok": not bool(bad.any()), "bit_exact": bool(np.array_equal(y, ref)), "max_rel_err": float(err.max()),
            "rows_checked": int(nrows), "rows_total": int(inc.N), "mismatches": int(bad.sum()),
            "bound": "1e-5*max(1,|ref|) + 2^-24 * (deg(v) + max|e|) for rows with chains longer than 64 terms",
            "against": "oracle (CPU restatement of the reference host path), same input"}


def run_config(shape, replicas, F, weighted, variant, steps, warmup, dev, sync, barrier, rank, opts_kw,
               want_cpu, want_parity, linear_out=0, linear_math="f32"):
    """Build one workload on `dev`, time `steps` aggregations, check the output.  Returns
    (result dict, cpu baseline dict, state for the extras).  linear_out > 0: the aggregation with the
    layer's bias-free linear F -> linear_out folded in (hg_aggr_linear_f32; reference
    model/ugsys/hgnn.py:22-23), the one MFMA contraction next to the path.  linear_math: 'f32' = fp32 MFMA; 'bf16x6' =
    each fp32 product of the matrix phase as six bf16 products (HG_LIN_BF16X6, include/hg_aggr.h) -- same checks, same
    bounds, and the measured error against float64 is on the line beside the fp32 form's."""
    import numpy as np
    import torch
    import hypergef_amd as hg
    from hypergef_amd import plan as planmod, synth

    base, inc = make_workload(shape, replicas)
    X_host = synth.features_like_reference(inc.N, F, seed=100 + rank)
    ptr = torch.from_numpy(inc.csrptr).to(dev)
    ind = torch.from_numpy(inc.colind).to(dev)
    X = torch.from_numpy(X_host).to(dev)
    opts = planmod.make_opts(**opts_kw)
    t0 = time.perf_counter()
    plan = planmod.Plan.from_tensors(inc.N, ptr, ind, opts)
    plan_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    resolved = plan.auto_variant(F) if variant == "auto" else variant
    fused_shape = plan.prepare(F) if resolved == "fused" else None
    prepare_s = time.perf_counter() - t0
    degE = degV = W = None
    scales_host = None
    n_w = 0
    if weighted:
        hyperg = hg.HyperGraph.from_incidence(inc, dev, ngs=1 << 30)
        degE, degV = hyperg.degE.reshape(-1), hyperg.degV.reshape(-1)
        degE = torch.where(torch.isinf(degE), torch.zeros_like(degE), degE)
        W = torch.ones(inc.M, device=dev)  # HGNNConv's Wdiag (model/ugsys/hgnn.py:12); random W: tests/test_gpu_parity.py
        scales_host = (degE.cpu().numpy(), degV.cpu().numpy(), W.cpu().numpy())
        n_w = 2
    weight = weight_host = None
    if linear_out:
        from hypergef_amd import _lib
        g = torch.Generator().manual_seed(7)
        weight_host = (torch.randn(linear_out, F, generator=g) / F ** 0.5).numpy()
        weight = torch.from_numpy(weight_host).to(dev)
        packed = planmod.pack_linear(weight, bf16x6=linear_math == "bf16x6")
        Y = torch.empty((inc.N, linear_out), dtype=torch.float32, device=dev)
        ws = torch.empty(max(int(_lib.lib().hg_aggr_linear_workspace_bytes(plan._h, F)), 256), dtype=torch.uint8, device=dev)

        def step():
            plan.aggregate_linear(ptr, ind, X, weight, degE, degV, W, variant=variant, out=Y, workspace=ws, packed=packed,
                                  math=linear_math)
    else:
        Y = torch.empty((inc.N, F), dtype=torch.float32, device=dev)
        ws = torch.empty(max(plan.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)

        def step():
            plan.aggregate(ptr, ind, X, degE, degV, W, variant=variant, out=Y, workspace=ws)

    for _ in range(warmup):
        step()
    wall, dev_s = timed_steps(step, steps, sync, barrier)

    launches = {"pull": 2, "fused": 1, "push_atomic": 1}[resolved]
    dominant = {"pull": "gather_rows_kernel (hop 1 + hop 2 launches averaged)",
                "fused": "fused_packed_kernel", "push_atomic": "push_tasks_kernel"}[resolved]
    info = fused_shape or {}
    helper_launches = 0
    if resolved == "fused":  # pre-pass / hub-pass / fixup launches of this schedule, timed inside the step
        helper_launches = int(info.get("n_mat", 0) > 0) + int(info.get("n_hub", 0) > 0) + int(info.get("fixups", 0) > 0)
    balg = b_alg(inc.N, inc.M, inc.nnz, F, n_w, degV is not None)
    if linear_out:  # Y is [N, linear_out]; the packed weight is read once
        balg += 4 * (inc.N * (linear_out - F) + linear_out * F)
        dominant = "fused_packed_kernel<..., LIN = true> (hop 1, hop 2, rows . Wlin^T on %s)" % (
            "v_mfma_f32_16x16x32_bf16, six bf16 products per fp32 product" if linear_math == "bf16x6" else "v_mfma_f32_16x16x4_f32")
    # the whole step's device time over its algorithmic bytes: helper launches count against the
    # step, so a schedule that needs them is not flattered
    step_s = dev_s / steps
    achieved = balg / step_s / 1e9
    name = workload_name(shape, replicas, F) + (", weighted (degE, degV, W)" if weighted else "")
    if linear_out:
        name += ", aggregation + linear %d->%d (hg_aggr_linear_f32%s)" % (F, linear_out, ", bf16x6" if linear_math == "bf16x6" else "")
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            entry = json.load(open(tpath)).get(name, {})
            traffic = entry.get("bytes_per_step", entry.get("bytes_per_launch"))
        except Exception:
            traffic = None
    res = {
        "workload": name, "short": short_name(shape, replicas, F, weighted, linear_out, linear_math),
        "op": ("(H*H^T*X) * Wlin^T (HGNNConv layer: aggregation + nn.Linear)" if linear_out else
                                 "hgnnaggr (degE, degV, W)" if weighted else "H*H^T*X (aggr_proto)"),
        "vertices": inc.N, "hyperedges": inc.M, "nnz": inc.nnz, "feat_len": F,
        "variant": variant, "resolved_variant": resolved,
        "ms_per_step": wall / steps * 1e3, "device_ms_per_step": step_s * 1e3,
        "edges_per_s": inc.nnz * steps / wall,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": dominant,
                     "algorithmic_bytes_per_step": balg, "avg_step_us": step_s * 1e6,
                     "algorithmic_bytes_per_launch": balg / launches, "avg_launch_us": step_s / launches * 1e6,
                     "launches_per_step": launches, "helper_launches_per_step": helper_launches},
        "plan_build_s": plan_s, "fused_schedule_build_s": prepare_s, "fused_schedule": fused_shape,
        "plan": {k: plan.info[k] for k in ("panels", "tasks", "fixups", "max_len", "short_max",
                                            "panel_rows", "panel_nnz")},
    }
    if linear_out:
        flops = 2.0 * inc.N * F * linear_out
        res["roofline_mfma"] = {"bound": "mfma", "achieved": flops / step_s / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": flops / step_s / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                "dtype": ("f32 as six bf16 products per product (v_mfma_f32_16x16x32_bf16, fp32 accumulate); flops = the "
                                          "fp32 contraction's 2 N K F_out, peak = the fp32 MFMA peak: an fp32-equivalent rate, "
                                          "not a bf16 pipe utilisation (that is 6 x flops / 2.5 PFLOP/s)" if linear_math == "bf16x6"
                                          else "f32 (v_mfma_f32_16x16x4_f32: exact fp32 in, fp32 accumulate)"),
                                "flops_per_step": flops, "traffic": None,
                                "time_at_peak_ms": flops / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3,
                                "time_at_hbm_peak_ms": balg / (HBM_PEAK_GBS * 1e9) * 1e3}
    cpu = None
    if rank == 0 and (want_cpu or want_parity):
        if linear_out:
            # the reference's order: project (float64 product rounded once to fp32), then the oracle's aggregation
            import numpy as np
            Z = (X_host.astype(np.float64) @ weight_host.T.astype(np.float64)).astype(np.float32)
            ref, nrows, cpu = oracle_pass(base, inc, linear_out, Z, weighted, scales_host, False)
        else:
            ref, nrows, cpu = oracle_pass(base, inc, F, X_host, weighted, scales_host, want_cpu)
        if want_parity:
            f64, mass = float64_report(Y, inc, X_host, scales_host, weight_host)
            if linear_out:
                # (A X) Wlin^T against A (X Wlin^T): the same fp32 fma chains in another order, each side within 1e-5 of
                # the element's l1 mass of the exact answer -> 2e-5 of it between them, per element
                import numpy as np
                err = np.abs(Y[:nrows].cpu().numpy() - ref) / mass[:nrows]
                res["parity"] = {"ok": bool(err.max() <= 2e-5), "bit_exact": False, "max_rel_err": float(err.max()),
                                 "rows_checked": int(nrows), "rows_total": int(inc.N), "mismatches": int((err > 2e-5).sum()),
                                 "bound": "2e-5*max(1, (|A||X||Wlin|^T)[v,k]) per element",
                                 "against": "oracle aggregation of the projected rows (linear-then-aggregate, the reference's order)"}
            else:
                res["parity"] = parity_report(Y, ref, nrows, inc)
            del mass
            res["parity"].update(f64)
            res["parity"]["ok"] = bool(res["parity"]["ok"] and f64["float64_ok"])
    state = dict(base=base, inc=inc, plan=plan, ptr=ptr, ind=ind, X=X, Y=Y, ws=ws, opts=opts,
                 degE=degE, degV=degV, W=W, wall=wall)
    return res, cpu, state


def single_graph_latency(state, F, dev, sync, shape):
    """Latency of ONE hypergraph (what result.xlsx "fig7,fig9" reports, ms per aggregation)."""
    import torch
    from hypergef_amd import plan as planmod
    base, X = state["base"], state["X"]
    p1 = torch.from_numpy(base.csrptr).to(dev)
    i1 = torch.from_numpy(base.colind).to(dev)
    pl1 = planmod.Plan.from_tensors(base.N, p1, i1, state["opts"])
    X1 = X[:base.N].contiguous()
    Y1 = torch.empty((base.N, F), dtype=torch.float32, device=dev)
    pl1.prepare(F)
    ws1 = torch.empty(max(pl1.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)
    single = {}
    for var in ("fused", "pull", "push_atomic"):
        def f():
            pl1.aggregate(p1, i1, X1, out=Y1, workspace=ws1, variant=var)
        for _ in range(20):
            f()
        g = torch.cuda.CUDAGraph()
        sync()
        with torch.cuda.graph(g):
            for _ in range(20):
                f()
        g.replay()
        _, d = timed_steps(g.replay, 20, sync, lambda: None)
        single[var + "_us"] = d / 400 * 1e6
    # what "auto" runs after the timed choice (hg_plan_tune_f32, the counterpart of the reference's tuner)
    tuned = pl1.tune(p1, i1, X1)

    def fa():
        pl1.aggregate(p1, i1, X1, out=Y1, workspace=ws1, variant="auto")
    for _ in range(20):
        fa()
    g = torch.cuda.CUDAGraph()
    sync()
    with torch.cuda.graph(g):
        for _ in range(20):
            fa()
    g.replay()
    _, d = timed_steps(g.replay, 20, sync, lambda: None)
    single["auto_tuned_us"] = d / 400 * 1e6
    single["tuned_choice"] = tuned["variant"] + ("" if tuned["variant"] == "fused" else "/hop kernels %d" % tuned["pull_hop_kernels"])
    single["reference_rtx3090_us"] = {"cora": 4.79, "citeseer": 3.70, "pubmed": 12.48}.get(shape)
    single["note"] = "device time per aggregation, 20 back-to-back aggregations per hipGraph replay"
    return single


def short_name(shape, replicas, F, weighted=False, linear_out=0, linear_math="f32"):
    """The name a configuration carries on the final line (the long one, `workload_name`, keys profiles/traffic.json)."""
    s = "powerlaw 1M/4M F=%d" % F if shape == "powerlaw" else "%s x%d F=%d" % (shape, replicas, F)
    if linear_out:
        s += " +linear%d (MFMA%s)" % (linear_out, " bf16x6" if linear_math == "bf16x6" else "")
    return s + (" weighted" if weighted else "")


def rocsparse_comparator(F, iters=100):
    """ONE cora / citeseer / pubmed-shape hypergraph each through the drop-in CLI (bin/aggr_proto, the reference's
    aggr_proto.cu flow): rocSPARSE two-step SpMM and this backend's best native variant, ms per aggregation, from the
    CSV row the CLI appends (aggr_proto.cu:51,82).  The reference quotes its result as a speedup over cuSPARSE's
    two-step SpMM on the same shapes; printed beside ours, with the absolute times: rocSPARSE has
    none of cuSPARSE's ~20 us fixed cost per call, so the ratio is not comparable -- the times are."""
    import csv
    import subprocess
    import tempfile
    from hypergef_amd import synth
    exe = os.path.join(ROOT, "bin", "aggr_proto")
    if not os.path.exists(exe):
        return {"error": "bin/aggr_proto not built (python -c 'import __graft_entry__ as g; g.build()')"}
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, make in (("cora", synth.cora_shape), ("citeseer", synth.citeseer_shape), ("pubmed", synth.pubmed_shape)):
            mtx = os.path.join(tmp, name + ".mtx")
            synth.write_mtx(mtx, make())
            try:
                r = subprocess.run([exe, mtx, str(F), "--iter", str(iters)], capture_output=True, text=True, cwd=tmp, timeout=120)
            except subprocess.TimeoutExpired:
                out[name] = {"error": "aggr_proto timed out"}
                continue
            ok = r.returncode == 0 and "check failed" not in r.stdout and "Wrong result" not in r.stdout
            rows = []
            if os.path.exists(os.path.join(tmp, "result.csv")):
                with open(os.path.join(tmp, "result.csv")) as f:
                    rows = [x for x in csv.reader(f) if x and x[0].endswith(name + ".mtx")]
            if not ok or not rows or not rows[-1][2] or not rows[-1][8]:
                out[name] = {"error": ("rc %d: " % r.returncode) + (r.stdout[-200:] + r.stderr[-200:]).strip()}
                continue
            two, native = float(rows[-1][2]) * 1e3, float(rows[-1][8]) * 1e3
            ref_two, ref_fused = (v * 1e3 for v in REFERENCE_RTX3090_F32[name])
            out[name] = {"native_us": native, "rocsparse_twostep_us": two, "speedup_over_rocsparse_twostep": two / native,
                         "reference_fused_us_rtx3090": ref_fused, "reference_cusparse_twostep_us_rtx3090": ref_two,
                         "reference_speedup_over_cusparse_twostep": ref_two / ref_fused}
    out["note"] = ("one hypergraph, F=%d, device us per aggregation via bin/aggr_proto; the reference's ratio is over cuSPARSE "
                   "(~20 us fixed cost per call) on an RTX 3090, ours over rocSPARSE on this GPU: compare the times" % F)
    return out


def sharded_section(args, dev, sync, barrier, rank, world):
    """ONE hypergraph hyperedge-partitioned over the ranks (hypergef_amd.dist.ShardedAggregator,
    default HIP operator per rank) with the dense exchange north_star names: sum all-reduce of the
    [N, F] partials (every rank ends with Y), and reduce-scatter (rank r keeps its row block: half
    the bytes).  Strong scaling: the work is fixed, so value = nnz / time.  An exchange that raises is
    recorded under its name AND fails the run (`errors` > 0 -> exit code 1): a broken RCCL path must
    not hide behind the headline."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from hypergef_amd import synth
    from hypergef_amd.dist import ShardedAggregator
    from hypergef_amd.plan import Plan
    F = args.sharded_feat
    inc = synth.powerlaw(args.sharded_nodes, args.sharded_edges, seed=3)
    X = torch.from_numpy(synth.features_like_reference(inc.N, F, seed=7)).to(dev)  # replicated: same seed
    out = {"workload": "power-law |V|=%d |E|=%d, F=%d, one hypergraph over %d ranks" % (inc.N, inc.M, F, world),
           "vertices": inc.N, "hyperedges": inc.M, "feat_len": F,
           "nnz": inc.nnz, "scaling": "strong", "rccl_ranks": world,
           "backend": dist.get_backend(), "bytes_per_rank_partial": inc.N * F * 4, "errors": 0}
    n = max(args.steps // 10, 5)
    full = None
    if rank == 0:  # single-GPU answer on rank 0's device: the sharded sum must reproduce it
        ptr = torch.from_numpy(inc.csrptr).to(dev)
        ind = torch.from_numpy(inc.colind).to(dev)
        plan1 = Plan.from_tensors(inc.N, ptr, ind)
        full = plan1.aggregate(ptr, ind, X)
        for _ in range(3):
            plan1.aggregate(ptr, ind, X, out=full)
        w1, _ = timed_steps(lambda: plan1.aggregate(ptr, ind, X, out=full), n, sync, lambda: None)
        out["single_gpu_ms"] = w1 / n * 1e3
        out["single_gpu_value"] = inc.nnz * n / w1
        del plan1
    for exchange in ("allreduce", "reduce_scatter", "allreduce_pipelined"):
        try:
            if args.inject_exchange_error and exchange == "reduce_scatter":
                raise RuntimeError("injected exchange error (test)")
            if exchange == "allreduce_pipelined":  # SURVEY 8(e) iv: column slices, collective c overlaps kernels c + 1
                agg = ShardedAggregator(inc, device=dev, exchange="allreduce", column_chunks=args.sharded_chunks,
                                        force_collective=args.force_collective)
            else:
                agg = ShardedAggregator(inc, device=dev, exchange=exchange, force_collective=args.force_collective)
            for _ in range(3):
                Y = agg.aggregate(X)
            w, _ = timed_steps(lambda: agg.aggregate(X), n, sync, barrier)
            t = torch.tensor([w], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = float(t.item())
            # the local kernel alone (no collective), for the split
            loc = ShardedAggregator(inc, device=dev, exchange="none")
            loc.aggregate(X)
            wl, _ = timed_steps(lambda: loc.aggregate(X), n, sync, barrier)
            entry = {"ms_per_step": w / n * 1e3, "value": inc.nnz * n / w, "unit": "edges/s",
                     "local_kernel_ms": wl / n * 1e3, "shard_hyperedges": [agg.lo, agg.hi]}
            if rank == 0:
                lo, hi = agg.row_range() if exchange == "reduce_scatter" else (0, inc.N)
                ref = full[lo:hi]
                err = ((Y - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
                entry["max_rel_err_vs_single_gpu"] = err
                entry["ok"] = bool(err <= 1e-5)
                if not entry["ok"]:
                    out["errors"] += 1
            if exchange == "allreduce_pipelined":
                entry["column_chunks"] = args.sharded_chunks
            out[exchange] = entry
        except Exception as exc:  # recorded and counted: the line is still printed, the run fails
            out[exchange] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
            out["errors"] += 1
    try:  # SURVEY 8(e) v: the whole hypergraph on every rank, F / world columns each, no collective
        from hypergef_amd.dist import ColumnShardedAggregator
        cols = ColumnShardedAggregator(inc, device=dev)
        c0, c1 = cols.columns(F)
        Xr = X[:, c0:c1].contiguous()
        for _ in range(3):
            Yr = cols.aggregate(Xr, sliced=True, F=F)
        w, _ = timed_steps(lambda: cols.aggregate(Xr, sliced=True, F=F), n, sync, barrier)
        t = torch.tensor([w], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        w = float(t.item())
        entry = {"ms_per_step": w / n * 1e3, "value": inc.nnz * n / w, "unit": "edges/s", "columns_per_rank": c1 - c0,
                 "note": "column-sharded: X and Y stay [N, F / ranks] per rank, no data-path collective"}
        if rank == 0:
            ref = full[:, c0:c1]
            err = ((Yr - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
            entry["max_rel_err_vs_single_gpu"] = err
            entry["ok"] = bool(err <= 1e-5)
            if not entry["ok"]:
                out["errors"] += 1
        out["column_sharded"] = entry
    except Exception as exc:
        out["column_sharded"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        out["errors"] += 1
    return out


# ------------------------------------------------------------------------------------------------------------------
# the final line

def _r(x, digits=4):
    """Numbers as they appear on the final line: 4 significant digits."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    return float("%.*g" % (digits, x))


def _config_record(r):
    """One configuration on the final line: {workload, ms_per_step, frac, traffic_over_algorithmic, parity_ok}
    (+ mfma_frac for the aggregation + linear entry; an entry that raised carries `error`)."""
    if "error" in r:
        return {"workload": r.get("short", r.get("workload", "?")), "error": str(r["error"])[:120], "parity_ok": False}
    roof = r["roofline"]
    rec = {"workload": r.get("short", r["workload"]), "ms_per_step": _r(r["device_ms_per_step"]), "frac": _r(roof["frac"]),
           "traffic_over_algorithmic": _r(roof["traffic"] / roof["algorithmic_bytes_per_step"], 3) if roof.get("traffic") else None,
           "parity_ok": r["parity"]["ok"] if "parity" in r else None}
    if "roofline_mfma" in r:
        rec["mfma_frac"] = _r(r["roofline_mfma"]["frac"])
        if "parity" in r:  # the two forms of the matrix phase side by side: error against float64, per element's mass
            rec["err_vs_f64"] = _r(r["parity"].get("max_rel_err_vs_float64"), 3)
    return rec


def compact_line(out):
    """The final stdout line from the full result `out` (what goes to bench_detail.json): one JSON object of
    at most LINE_LIMIT bytes.  Optional parts are dropped, least important first, if it would not fit."""
    roof = out["roofline"]
    line = {k: _r(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = out["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "op", "feat_len", "variant", "resolved_variant", "sharding") if k in cfg}
    line["roofline"] = {k: _r(roof.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                                                      "algorithmic_bytes_per_step", "avg_step_us", "launches_per_step",
                                                      "helper_launches_per_step")}
    line["hbm_gbs_algorithmic"] = _r(out.get("hbm_gbs_algorithmic"))
    cpu = out.get("cpu_baseline")
    if cpu:
        c = {"value": _r(cpu["value"]), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"], "sample": cpu["sample"][:140]}
        if "all_cores" in cpu:
            c["all_cores"] = {"value": _r(cpu["all_cores"]["value"]), "cores": cpu["all_cores"]["cores"],
                              "note": "shared host, informative; the 1-core figure is the stable baseline"}
        c["host"] = cpu.get("host")
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    par = out.get("parity")
    records = [_config_record(r) for r in out.get("configs_detail", [])]
    if par:
        line["parity"] = {"ok": par["ok"], "max_rel_err": _r(par["max_rel_err"]),
                          "max_rel_err_vs_float64": _r(par.get("max_rel_err_vs_float64")),
                          "all_configs_ok": bool(par["ok"] and all(r.get("parity_ok") in (True, None) and "error" not in r
                                                                   for r in records))}
    else:
        line["parity"] = None
    line["configs"] = records
    if out.get("device_copy"):
        line["device_copy_gbs"] = _r(out["device_copy"]["gbs"])
    comp = out.get("comparator")
    if comp:
        sg = {}
        for name in ("cora", "citeseer", "pubmed"):
            e = comp.get(name)
            if e and "error" not in e:
                sg[name] = {"us": _r(e["native_us"], 3), "rocsparse_twostep_us": _r(e["rocsparse_twostep_us"], 3),
                            "speedup_over_rocsparse_twostep": _r(e["speedup_over_rocsparse_twostep"], 3),
                            "reference_us_rtx3090": _r(e["reference_fused_us_rtx3090"], 3),
                            "reference_speedup_over_cusparse": _r(e["reference_speedup_over_cusparse_twostep"], 3)}
            elif e:
                sg[name] = {"error": e["error"][:80]}
        if "error" in comp:
            sg["error"] = comp["error"][:120]
        sg["note"] = "one hypergraph F=32, us/aggregation; reference ratio is vs cuSPARSE (~20us/call fixed cost): compare times"
        line["single_graph"] = sg
    sh = out.get("sharded")
    if sh:
        ss = {"workload": "powerlaw |V|=%s |E|=%s F=%s, one hypergraph, hyperedge-sharded" % (
                  sh.get("vertices", "1M"), sh.get("hyperedges", "4M"), sh.get("feat_len", 64)),
              "rccl_ranks": sh["rccl_ranks"], "backend": sh["backend"], "unit": "edges/s",
              "single_gpu_ms": _r(sh.get("single_gpu_ms")), "errors": sh.get("errors", 0)}
        for form in ("allreduce", "reduce_scatter", "allreduce_pipelined", "column_sharded"):
            e = sh.get(form)
            if not e:
                continue
            if "error" in e:
                ss[form] = {"error": e["error"][:100]}
            else:
                ss[form] = {"value": _r(e["value"]), "ms_per_step": _r(e["ms_per_step"]),
                            "local_kernel_ms": _r(e.get("local_kernel_ms")),
                            "max_rel_err_vs_single_gpu": _r(e.get("max_rel_err_vs_single_gpu"), 2)}
        line["strong_scaling"] = ss
        line["scaling_note"] = "value/scaling: independent batches per rank, no collective (weak); strong_scaling: north_star's sharded form"
    line["detail"] = os.path.basename(out.get("detail_path", "bench_detail.json"))
    for drop in (None, ("scaling_note",), ("hbm_gbs_algorithmic", "device_copy_gbs"), ("single_graph",)):
        for k in drop or ():
            line.pop(k, None)
        s = json.dumps(line, separators=(",", ":"))
        if len(s.encode()) < LINE_LIMIT:
            return s
    # last resort: shorten strings (cannot happen with the fixed set of configurations; kept so the limit is a guarantee)
    line["config"] = {"workload": cfg["workload"][:60]}
    line["roofline"]["kernel"] = str(line["roofline"]["kernel"])[:40]
    if line.get("cpu_baseline"):
        line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:40]
    for rec in line["configs"]:
        rec["workload"] = rec["workload"][:24]
        rec.pop("error", None)
    s = json.dumps(line, separators=(",", ":"))
    while len(s.encode()) >= LINE_LIMIT and len(line["configs"]) > 1:
        line["configs"].pop()
        s = json.dumps(line, separators=(",", ":"))
    return s


def write_detail(out, path):
    """The full result as indented JSON: next to bench.py (or --detail) and, when it exists, under gpurun_out/."""
    written = []
    targets = [path]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        targets.append(os.path.join(ROOT, "gpurun_out", os.path.basename(path)))
    for p in targets:
        try:
            with open(p, "w") as f:
                json.dump(out, f, indent=1)
            written.append(p)
        except OSError as exc:
            print("bench.py: could not write %s: %s" % (p, exc), file=sys.stderr)
    return written


# ------------------------------------------------------------------------------------------------------------------

def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start `python -m torch.distributed.run --nproc-per-node N`
    on this file as a CHILD process -- nothing in this process has touched the GPU yet, and it never will -- let it
    write to our stdout / stderr (rank 0's final line is then the last line of ours) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    print("bench.py: --gpus %d outside torchrun: launching %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def rehearse_config(args, rank):
    """--rehearse-cpu: a stand-in for run_config with the same result shape and NO kernels (a sleep per step), so the
    launch / rendezvous / max-over-ranks / output-line control flow runs where there is no GPU."""
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.0005 * (rank + 1))
    wall = time.perf_counter() - t0
    name = workload_name(args.shape, args.replicas, args.feat)
    nnz, N, M = 4859 * args.replicas, 2708 * args.replicas, 1579 * args.replicas
    balg = b_alg(N, M, nnz, args.feat, 0, False)
    step_s = wall / args.steps
    res = {"workload": name, "short": short_name(args.shape, args.replicas, args.feat), "op": "rehearsal: no kernel ran",
           "vertices": N, "hyperedges": M, "nnz": nnz, "feat_len": args.feat, "variant": args.variant,
           "resolved_variant": "none", "ms_per_step": step_s * 1e3, "device_ms_per_step": step_s * 1e3,
           "edges_per_s": nnz / step_s,
           "roofline": {"bound": "hbm", "achieved": balg / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": balg / step_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "none (rehearsal)",
                        "algorithmic_bytes_per_step": balg, "avg_step_us": step_s * 1e6, "launches_per_step": 0,
                        "helper_launches_per_step": 0},
           "plan_build_s": 0.0, "fused_schedule_build_s": 0.0, "fused_schedule": None, "plan": None}

    class _Inc:
        pass
    inc = _Inc()
    inc.N, inc.M, inc.nnz = N, M, nnz
    return res, None, {"inc": inc, "wall": wall}


def rehearse_sharded(args, rank, world):
    """--rehearse-cpu: the exchange step of the `sharded` section on CPU tensors over gloo (a real all-reduce and a
    real reduce-scatter-by-all-reduce whose sums are checked), same result shape, same error accounting."""
    import torch
    import torch.distributed as dist
    out = {"workload": "rehearsal", "nnz": 1000, "scaling": "strong", "rccl_ranks": world, "backend": dist.get_backend(),
           "errors": 0, "single_gpu_ms": 1.0}
    for exchange in ("allreduce", "reduce_scatter"):
        try:
            if args.inject_exchange_error and exchange == "reduce_scatter":
                raise RuntimeError("injected exchange error (test)")
            part = torch.full((64, 8), float(rank + 1))
            t0 = time.perf_counter()
            dist.all_reduce(part)
            w = time.perf_counter() - t0
            err = float((part - world * (world + 1) / 2).abs().max())
            out[exchange] = {"ms_per_step": w * 1e3, "value": 1000 / max(w, 1e-9), "unit": "edges/s", "local_kernel_ms": 0.0,
                             "max_rel_err_vs_single_gpu": err, "ok": err == 0.0}
            out["errors"] += int(err != 0.0)
        except Exception as exc:
            out[exchange] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
            out["errors"] += 1
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:  # rehearsal on a one-GPU box: every rank on cuda:0, collectives over gloo
        local_rank = 0
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rehearse = args.rehearse_cpu
    if rehearse:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.share_gpu or rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    def sync():
        if not rehearse:
            torch.cuda.synchronize(dev)

    opts_kw = dict(short_max=args.short_max, panel_rows=args.panel_rows, panel_nnz=args.panel_nnz,
                   xcd_remap=not args.no_xcd_remap, t_big=args.t_big, fused_tile_bytes=args.tile_bytes,
                   fused_steps=args.fused_steps, hub_pass=not args.no_hub_pass,
                   row_stream=not args.no_row_stream)
    F = args.feat
    one = world == 1
    if rehearse:
        barrier()
        res, cpu, st = rehearse_config(args, rank)
        barrier()
    else:
        res, cpu, st = run_config(args.shape, args.replicas, F, args.weighted, args.variant, args.steps, args.warmup,
                                  dev, sync, barrier, rank, opts_kw,
                                  want_cpu=one and not args.no_cpu_baseline, want_parity=not args.no_parity,
                                  linear_out=args.linear_out, linear_math=args.linear_math)  # N > 1: rank 0 still checks its own shard's timed output
    wall = st["wall"]
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    inc = st["inc"]
    total_nnz = inc.nnz * world
    value = total_nnz * args.steps / wall
    out = {
        "metric": "aggregated edges/sec (fused V->E->V aggregation)",
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not rehearse else "rehearsal (no kernels run)",
        "config": {"workload": res["workload"], "op": res["op"],
                   "vertices_per_gpu": inc.N, "hyperedges_per_gpu": inc.M, "nnz_per_gpu": inc.nnz,
                   "feat_len": F, "variant": args.variant, "resolved_variant": res["resolved_variant"],
                   "sharding": "hyperedge groups (one hypergraph batch per rank), no data-path collective"},
        "roofline": res["roofline"],
        "hbm_gbs_algorithmic": res["roofline"]["algorithmic_bytes_per_step"] * world * args.steps / wall / 1e9,
        "plan_build_s": res["plan_build_s"], "fused_schedule_build_s": res["fused_schedule_build_s"],
        "fused_schedule": res["fused_schedule"], "plan": res["plan"], "detail_path": args.detail,
    }
    if "parity" in res:
        out["parity"] = res["parity"]
    failed = "parity" in res and not res["parity"]["ok"]

    if not args.no_extras and not args.linear_out and not rehearse:
        # What a plain device copy X -> Y (the 2NF term of B_alg, no gather, no index traffic)
        # takes on this box: the practical floor of any kernel that reads X and writes Y once.
        X, Y = st["X"], st["Y"]
        for _ in range(5):
            Y.copy_(X)
        _, d = timed_steps(lambda: Y.copy_(X), 50, sync, lambda: None)
        copy_s = d / 50
        out["device_copy"] = {"ms": copy_s * 1e3, "gbs": 2.0 * inc.N * F * 4 / copy_s / 1e9,
                              "step_over_copy": (wall / args.steps) / copy_s,
                              "note": "torch copy of X into Y, same buffers; step_over_copy = "
                                      "aggregation step time / this"}
        if args.shape != "powerlaw":
            out["single_graph"] = single_graph_latency(st, F, dev, sync, args.shape)
        copy_gbs = out["device_copy"]["gbs"]
        floor_of(out["roofline"], copy_gbs)
    del st
    if not rehearse:
        torch.cuda.empty_cache()

    if (world > 1 or args.force_collective) and not args.no_extras:
        out["sharded"] = rehearse_sharded(args, rank, world) if rehearse else sharded_section(args, dev, sync, barrier, rank, world)
        errs = torch.tensor([out["sharded"]["errors"]], dtype=torch.int64, device=dev)
        dist.all_reduce(errs, op=dist.ReduceOp.MAX)  # any rank's failure fails every rank
        out["sharded"]["errors"] = int(errs.item())
        failed = failed or out["sharded"]["errors"] > 0

    # every configuration of the line, the headline's own cell first
    head = dict(res)
    head.setdefault("short", short_name(args.shape, args.replicas, F, args.weighted, args.linear_out, args.linear_math))
    for k in ("plan", "plan_build_s", "fused_schedule"):
        head.pop(k, None)
    configs = [head]
    if one and not args.no_configs and not args.no_extras and not rehearse:
        # the other BASELINE configurations, same measurement, bounded step counts: north_star's target matrix (the
        # headline is its cora F = 32 cell; replica counts put X + Y beyond the 256 MiB Infinity Cache), config 3 and
        # its MFMA path, config 4, the weighted operator
        todo = [("citeseer", 1024, 32, False, 0), ("pubmed", 256, 32, False, 0), ("cora", 256, 128, False, 0),
                ("citeseer", 256, 128, False, 0), ("pubmed", 64, 128, False, 0), ("pubmed", 64, 128, False, 128),
                ("pubmed", 64, 128, False, 128, "bf16x6"),
                ("powerlaw", 1, 64, False, 0), (args.shape, args.replicas, F, True, 0)]
        for shape, reps, feat, weighted, lin, *rest in todo:
            lmath = rest[0] if rest else "f32"
            if (shape, reps, feat, weighted, lin, lmath) == (args.shape, args.replicas, F, args.weighted, args.linear_out, args.linear_math):
                continue
            try:
                r, c, s2 = run_config(shape, reps, feat, weighted, "auto", args.config_steps, 10, dev, sync, barrier,
                                      rank, dict(xcd_remap=True), want_cpu=False, want_parity=not args.no_parity,
                                      linear_out=lin, linear_math=lmath)
                del s2
                torch.cuda.empty_cache()
                for k in ("plan", "plan_build_s"):
                    r.pop(k, None)
                failed = failed or ("parity" in r and not r["parity"]["ok"])
                if "device_copy" in out:
                    floor_of(r["roofline"], out["device_copy"]["gbs"])
                configs.append(r)
            except Exception as exc:  # recorded on the line (parity_ok false) and the run fails
                configs.append({"workload": workload_name(shape, reps, feat), "short": short_name(shape, reps, feat, weighted, lin, lmath),
                                "error": "%s: %s" % (type(exc).__name__, str(exc)[:300])})
                failed = True
    out["configs_detail"] = configs
    if one and not args.no_extras and not args.no_comparator and not rehearse and rank == 0:
        out["comparator"] = rocsparse_comparator(32)

    if rank == 0:
        out["cpu_baseline"] = cpu if one and not args.no_cpu_baseline else None
        line = compact_line(out)
        write_detail(out, args.detail)
        print(line, flush=True)
    if world > 1 or args.force_collective:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        print("bench.py: a timed output does not match the oracle, a configuration raised, or an exchange of the "
              "sharded section failed (see `parity`, `configs`, `strong_scaling`)", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
