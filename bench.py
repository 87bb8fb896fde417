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

N > 1: one process per GPU (torchrun); the batch is sharded by hyperedge group
= by hypergraph, K graphs per rank (weak scaling).  No vertex is shared between
shards, so the data path has no collective; ranks meet in the barrier that
brackets the timed region.  The dense all-reduce variant (every rank ends with
all of Y) is timed separately and reported under `allreduce_dense`.

One JSON line on stdout (rank 0).
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


def b_alg(N, M, nnz, F, n_w):
    """Algorithmic (compulsory) bytes of one fused aggregation, SURVEY.md 8(d)."""
    return 4 * (2 * N * F + 2 * nnz + (M + 1) + (N + 1) + n_w * M + N)


def parse():
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
    p.add_argument("--weighted", action="store_true", help="hgnnaggr (degE, degV, W) instead of H H^T X")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip single-graph / all-reduce extras")
    p.add_argument("--short-max", type=int, default=0)
    p.add_argument("--panel-rows", type=int, default=0)
    p.add_argument("--panel-nnz", type=int, default=0)
    p.add_argument("--no-xcd-remap", action="store_true")
    p.add_argument("--share-gpu", action="store_true",
                   help="rehearse the N>1 path on one GPU: all ranks use cuda:0, gloo instead of RCCL")
    return p.parse_args()


def make_workload(args, rank):
    from hypergef_amd import synth
    if args.shape == "powerlaw":
        base = synth.powerlaw(1_000_000, 4_000_000, seed=3)
        inc = base
    else:
        base = {"cora": synth.cora_shape, "citeseer": synth.citeseer_shape,
                "pubmed": synth.pubmed_shape}[args.shape]()
        inc = synth.replicate_block_diagonal(base, args.replicas)
    return base, inc


def timed_steps(fn, steps, sync, barrier):
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


def cpu_baseline(base, inc, F, X_host):
    """The oracle (a port of util::hyperaggr_reference_host, check.cuh:83-114)
    on this box's host cores, rank 0 only, on a bounded sample."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    # sample: the first S hypergraphs of the batch (whole workload if small)
    blocks = inc.M // base.M
    if blocks == 1 and inc.nnz > 2_000_000:
        # one big hypergraph (power-law config): the fused host path costs sum_v sum_e |e| row
        # reads, minutes at this size; time the reference's other CPU path, two spmm_reference_host
        # calls (spmm.cuh:724-740), once over the whole graph
        H_ptr, H_ind = orc.transpose_csr(inc.M, inc.N, inc.csrptr, inc.colind)
        t0 = time.perf_counter()
        orc.twostep_host(inc.N, inc.M, F, H_ptr, H_ind, inc.csrptr, inc.colind, X_host)
        dt = time.perf_counter() - t0
        return {"value": inc.nnz / dt, "unit": "edges/s", "cores": 1, "kind": "port",
                "sample": "whole %s hypergraph, F=%d, one pass of the two-step host path (%.2f s)"
                          % (base.name, F, dt)}
    S = min(blocks, 1024)
    Ms, Ns = base.M * S, base.N * S
    ptr = inc.csrptr[:Ms + 1]
    ind = inc.colind[:ptr[-1]]
    H_ptr, H_ind = orc.transpose_csr(Ms, Ns, ptr, ind)
    Xs = np.ascontiguousarray(X_host[:Ns])
    nnz_s = int(ptr[-1])
    best, passes = None, 0
    t_all = time.perf_counter()
    while passes < 3 or (time.perf_counter() - t_all < 10 and passes < 50):  # about 10 s of CPU work
        t0 = time.perf_counter()
        orc.hyperaggr_host(Ns, F, H_ptr, H_ind, ptr, ind, Xs)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        passes += 1
        if time.perf_counter() - t_all > 30:
            break
    out = {"value": nnz_s / best, "unit": "edges/s", "cores": 1, "kind": "port",
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
        out["all_cores"] = {"value": nnz_s / best_mt, "cores": threads}
        out["host"] = "%d logical cpus" % (os.cpu_count() or 0)
    except Exception as exc:  # the baseline is informative; never fail the bench on it
        out["all_cores_error"] = str(exc)
    return out


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:  # rehearsal on a one-GPU box: every rank on cuda:0, collectives over gloo
        local_rank = 0
    if world != max(args.gpus, 1):
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d"
                             % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    def sync():
        torch.cuda.synchronize(dev)

    import hypergef_amd as hg
    from hypergef_amd import plan as planmod, synth

    base, inc = make_workload(args, rank)
    F = args.feat
    X_host = synth.features_like_reference(inc.N, F, seed=100 + rank)
    ptr = torch.from_numpy(inc.csrptr).to(dev)
    ind = torch.from_numpy(inc.colind).to(dev)
    X = torch.from_numpy(X_host).to(dev)
    opts = planmod.make_opts(short_max=args.short_max, panel_rows=args.panel_rows,
                             panel_nnz=args.panel_nnz, xcd_remap=not args.no_xcd_remap,
                             t_big=args.t_big, fused_tile_bytes=args.tile_bytes)
    t0 = time.perf_counter()
    plan = planmod.Plan.from_tensors(inc.N, ptr, ind, opts)
    plan_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    fused_shape = plan.prepare(F) if plan.auto_variant(F) == "fused" or args.variant == "fused" else None
    prepare_s = time.perf_counter() - t0
    degE = degV = W = None
    n_w = 0
    if args.weighted:
        hyperg = hg.HyperGraph.from_incidence(inc, dev, ngs=1 << 30)
        degE, degV = hyperg.degE.reshape(-1), hyperg.degV.reshape(-1)
        W = torch.ones(inc.M, device=dev)
        n_w = 2
    Y = torch.empty((inc.N, F), dtype=torch.float32, device=dev)
    ws = torch.empty(max(plan.workspace_bytes(F), 256), dtype=torch.uint8, device=dev)

    def step():
        plan.aggregate(ptr, ind, X, degE, degV, W, variant=args.variant, out=Y, workspace=ws)

    for _ in range(args.warmup):
        step()
    wall, dev_s = timed_steps(step, args.steps, sync, barrier)
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    total_nnz = inc.nnz * world
    value = total_nnz * args.steps / wall
    resolved = plan.auto_variant(F) if args.variant == "auto" else args.variant
    launches = 2 if resolved == "pull" else 1
    dominant = {"pull": "gather_rows_kernel (hop 1 + hop 2 launches averaged)",
                "fused": "fused_packed_kernel", "push_atomic": "push_groups_kernel"}[resolved]
    balg = b_alg(inc.N, inc.M, inc.nnz, F, n_w)
    kern_avg_s = dev_s / (args.steps * launches)
    achieved = balg / launches / kern_avg_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    wl_name = "%s-shape x%d block-diagonal batch, F=%d" % (args.shape, args.replicas, F) \
        if args.shape != "powerlaw" else "power-law |V|=1M |E|=4M, F=%d" % F
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(wl_name, {}).get("bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "aggregated edges/sec (fused V->E->V aggregation)",
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl_name, "op": "hgnnaggr" if args.weighted else "H*H^T*X (aggr_proto)",
                   "vertices_per_gpu": inc.N, "hyperedges_per_gpu": inc.M, "nnz_per_gpu": inc.nnz,
                   "feat_len": F, "variant": args.variant, "resolved_variant": resolved,
                   "sharding": "hyperedge groups (one hypergraph batch per rank), no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": dominant,
                     "algorithmic_bytes_per_launch": balg / launches,
                     "avg_launch_us": kern_avg_s * 1e6, "launches_per_step": launches},
        "hbm_gbs_algorithmic": balg * world * args.steps / wall / 1e9,
        "plan_build_s": plan_s, "fused_schedule_build_s": prepare_s, "fused_schedule": fused_shape,
        "plan": {k: plan.info[k] for k in ("panels", "tasks", "fixups", "max_len", "short_max",
                                            "panel_rows", "panel_nnz")},
    }

    if not args.no_extras:
        # What a plain device copy X -> Y (the 2NF term of B_alg, no gather, no index traffic)
        # takes on this box: the practical floor of any kernel that reads X and writes Y once.
        for _ in range(5):
            Y.copy_(X)
        _, d = timed_steps(lambda: Y.copy_(X), 50, sync, lambda: None)
        copy_s = d / 50
        out["device_copy"] = {"ms": copy_s * 1e3, "gbs": 2.0 * inc.N * F * 4 / copy_s / 1e9,
                              "step_over_copy": (wall / args.steps) / copy_s,
                              "note": "torch copy of X into Y, same buffers; step_over_copy = "
                                      "aggregation step time / this"}
        # latency of ONE hypergraph (what result.xlsx "fig7,fig9" reports, ms per aggregation)
        if args.shape != "powerlaw":
            p1 = torch.from_numpy(base.csrptr).to(dev)
            i1 = torch.from_numpy(base.colind).to(dev)
            pl1 = planmod.Plan.from_tensors(base.N, p1, i1, opts)
            X1 = X[:base.N].contiguous()
            Y1 = torch.empty((base.N, F), dtype=torch.float32, device=dev)
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
            single["reference_rtx3090_us"] = {"cora": 4.79, "citeseer": 3.70, "pubmed": 12.48}.get(args.shape)
            single["note"] = "device time per aggregation, 20 back-to-back aggregations per hipGraph replay"
            out["single_graph"] = single
        if world > 1:
            # dense variant: every rank ends with the full Y of the global batch
            Yg = torch.zeros((inc.N * world, F), dtype=torch.float32, device=dev)

            def step_ar():
                plan.aggregate(ptr, ind, X, degE, degV, W, variant=args.variant, out=Y, workspace=ws)
                Yg[rank * inc.N:(rank + 1) * inc.N].copy_(Y)
                dist.all_reduce(Yg)
            for _ in range(3):
                step_ar()
            w_ar, _ = timed_steps(step_ar, max(args.steps // 10, 5), sync, barrier)
            t = torch.tensor([w_ar], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n_ar = max(args.steps // 10, 5)
            out["allreduce_dense"] = {"ms_per_step": float(t.item()) / n_ar * 1e3,
                                      "value": total_nnz * n_ar / float(t.item()),
                                      "bytes_allreduced": inc.N * world * F * 4}
            if not args.share_gpu:  # gloo has no reduce_scatter
                # SURVEY 8(e) option ii: the sum scattered, each rank keeps 1/world of the rows
                Yrs = torch.empty((inc.N, F), dtype=torch.float32, device=dev)

                def step_rs():
                    plan.aggregate(ptr, ind, X, degE, degV, W, variant=args.variant, out=Y, workspace=ws)
                    Yg[rank * inc.N:(rank + 1) * inc.N].copy_(Y)
                    dist.reduce_scatter_tensor(Yrs, Yg)
                try:  # an extra: never let it take the headline line down with it
                    for _ in range(3):
                        step_rs()
                    w_rs, _ = timed_steps(step_rs, n_ar, sync, barrier)
                    t = torch.tensor([w_rs], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    out["reduce_scatter_dense"] = {"ms_per_step": float(t.item()) / n_ar * 1e3,
                                                   "value": total_nnz * n_ar / float(t.item())}
                except Exception as exc:
                    out["reduce_scatter_dense"] = {"error": str(exc)[:200]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(base, inc, F, X_host)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
