#!/usr/bin/env python3
"""Training / inference timing driver, the reference's `hgsys.py` (which its README
calls `ugsys.py`) on this backend: same flags (`--model` or `--model-name`, `--backend`,
`--dname`, `--nhid`, `--nlayer`, `--nhead`, `--first-aggr`, `--epochs`, `--device`,
`--output`, `--profile`), same loop (10 warm-up epochs, `epochs` timed training epochs
with Adam(lr=0.01, wd=5e-4) and nll_loss, then `epochs` timed eval passes,
HyperGsys/hgsys.py:136-211), same CSV line.  Datasets are synthetic:
`--dname cora|citeseer|pubmed` selects the shape; features / labels are seeded random.
Backends: `hgsys` (this backend's kernels) and `torch` (index_add_ baseline standing
in for PyG/DGL, runs on `--device cpu` too).  `--world-size N` under torchrun shards a
batch of `--replicas` hypergraphs by hypergraph across ranks (data parallel: gradients
are all-reduced over RCCL).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--dname", default="cora")
    p.add_argument("--model", "--model-name", dest="model", type=str, default="HGNN", help="HGNN | UniGIN | UniGCNII")
    p.add_argument("--activation", type=str, default="relu")
    p.add_argument("--nlayer", type=int, default=2)
    p.add_argument("--nhid", type=int, default=32)
    p.add_argument("--nhead", type=int, default=1)
    p.add_argument("--nfeat", type=int, default=64, help="synthetic input feature width")
    p.add_argument("--nclass", type=int, default=7)
    p.add_argument("--dropout", type=float, default=0.6)
    p.add_argument("--input-drop", type=float, default=0.6)
    p.add_argument("--first-aggr", type=str, default="sum")
    p.add_argument("--lr", type=float, default=0.01)
    p.add_argument("--wd", type=float, default=5e-4)
    p.add_argument("--backend", type=str, default="hgsys", help="hgsys | torch")
    p.add_argument("--epochs", type=int, default=200)
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--replicas", type=int, default=1, help="hypergraphs in the (block-diagonal) batch")
    p.add_argument("--train_prop", type=float, default=0.5)
    p.add_argument("--profile", type=int, default=0)
    p.add_argument("--graph", action="store_true",
                   help="inference as one hipGraph replay per forward (launch-bound models: a dataset-sized hypergraph)")
    p.add_argument("--graph-train", action="store_true",
                   help="the training step (forward, loss, backward, Adam) as one hipGraph replay per epoch")
    p.add_argument("--no-graph", action="store_true",
                   help="never capture.  Default: a model over ONE launch-bound hypergraph (at most 2^18 incidences, one "
                        "GPU) runs its training step and its forward as hipGraph replays, both backends alike -- eager, "
                        "such an epoch is ~100 launches of host latency; the eager figures are printed beside the replays")
    p.add_argument("--linear-math", default="f32", choices=["f32", "bf16x6"],
                   help="hgsys backend, layers fused at nhid = 128 on batches: fp32 MFMA, or six bf16 products per fp32 product")
    p.add_argument("--output", type=str, default=None)
    return p.parse_args()


def main():
    args = parse()
    import torch.distributed as dist
    import hypergef_amd as hg
    from hypergef_amd import models, ops, synth
    ops.set_linear_math(args.linear_math)  # process default: the models resolve their options at every forward

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if args.device.startswith("cuda"):
            args.device = "cuda:%d" % local
            torch.cuda.set_device(local)
        dist.init_process_group("nccl" if args.device.startswith("cuda") else "gloo")
    torch.manual_seed(args.seed + rank)
    np.random.seed(args.seed + rank)
    dev = torch.device(args.device)

    base = {"cora": synth.cora_shape, "citeseer": synth.citeseer_shape, "pubmed": synth.pubmed_shape}[args.dname]()
    inc = synth.replicate_block_diagonal(base, args.replicas)
    hyperg = hg.HyperGraph.from_incidence(inc, dev, data_name=args.dname)
    X = torch.randn(inc.N, args.nfeat, device=dev)
    y = torch.randint(0, args.nclass, (inc.N,), device=dev)
    perm = torch.randperm(inc.N, device=dev)
    train_idx = perm[: int(args.train_prop * inc.N)]

    if args.model != "UniGCNII":
        model = models.HGsysHGNN(args, hyperg, args.nfeat, args.nhid, args.nclass, args.nlayer,
                                 args.first_aggr, args.nhead)
    else:
        model = models.UniGCNII(args, hyperg, args.nfeat, args.nhid, args.nclass, args.nlayer, args.nhead)
    model.to(dev)
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(
            model, device_ids=[dev.index] if dev.type == "cuda" else None)
    # One dataset-sized hypergraph is launch-bound end to end (HGNN on cora: a forward is 102 us of mostly host launch
    # latency, 26 us as a replay): capture is the default there, for both backends; --no-graph opts out.
    launch_bound = dev.type == "cuda" and world == 1 and inc.nnz <= (1 << 18) and not args.no_graph and not args.profile
    graph_train = dev.type == "cuda" and world == 1 and not args.no_graph and (args.graph_train or launch_bound)
    graph_infer = dev.type == "cuda" and not args.no_graph and (args.graph or launch_bound)
    # the eager epochs run the plain Adam (what a run without capture runs); the captured section gets a capturable
    # twin (step counters on the device) carrying the same state
    opti = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.wd)
    if rank == 0:
        print(f"Total Epochs: {args.epochs}")
        print(f"total_params:{sum(p.numel() for p in model.parameters() if p.requires_grad)}")

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def train_epoch():
        model.train()
        opti.zero_grad()
        Z = model(X)
        loss = F.nll_loss(Z[train_idx], y[train_idx])
        loss.backward()
        opti.step()
        # detached: a loss that keeps its graph alive also keeps the parameters' AccumulateGrad nodes alive, with
        # the stream they were created on -- a later capture on another stream would then accumulate gradients on
        # that (non-capturing) stream and the capture breaks
        return loss.detach()

    for _ in range(10):
        loss = train_epoch()
    sync()
    start = time.time()
    for _ in range(args.epochs):
        loss = train_epoch()
    sync()
    trainTime = (time.time() - start) / args.epochs
    if graph_train:
        # One dataset-sized hypergraph: an epoch is ~100 launch-bound kernels and the host's launch latency is most
        # of it.  Record the whole step once and replay it.  Everything a call allocates or decides on first use
        # (plans, bound scale sets, the all-ones test of Wdiag) exists after the eager epochs above; caches keyed
        # on torch's version counters are emptied around the capture, because a replay rewrites the weights behind
        # those counters (a packed weight cached under (address, version) would go stale).
        from hypergef_amd import plan as planmod
        eager_time = trainTime
        # the captured section trains on (warm-up, capture, timed replays): snapshot what the eager epochs produced, so
        # that the evaluation below sees the model after exactly 10 + epochs steps, as in a run without capture
        snap = [p.detach().clone() for p in model.parameters()]
        eager_opti, eager_loss = opti, loss
        try:
            sd = eager_opti.state_dict()
            for grp in sd["param_groups"]:
                grp["capturable"] = True
            opti = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.wd, capturable=True)
            opti.load_state_dict(sd)
            planmod.clear_pack_cache()
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    train_epoch()
            torch.cuda.current_stream(dev).wait_stream(side)
            planmod.clear_pack_cache()
            g = torch.cuda.CUDAGraph()
            opti.zero_grad(set_to_none=True)
            model.train()
            with torch.cuda.graph(g):
                Zs = model(X)
                static_loss = F.nll_loss(Zs[train_idx], y[train_idx])
                static_loss.backward()
                opti.step()
            planmod.clear_pack_cache()
            for _ in range(10):
                g.replay()
            sync()
            start = time.time()
            for _ in range(args.epochs):
                g.replay()
            sync()
            trainTime = (time.time() - start) / args.epochs
            replay_loss = static_loss.detach().clone()
            assert torch.isfinite(replay_loss), "training diverged under replay"
            if rank == 0:
                print(f"backend {args.backend}: avg epoch time {trainTime:.6f} as a hipGraph replay (eager, plain Adam: "
                      f"{eager_time:.6f}), loss after the replays {replay_loss.item():.4f}")
        except Exception as exc:
            sync()
            if rank == 0:
                print("graph capture of the training step failed (%s: %s); eager figure kept" % (type(exc).__name__, str(exc)[:200]))
        finally:
            sync()
            with torch.no_grad():
                for p, q in zip(model.parameters(), snap):
                    p.copy_(q)
            opti, loss = eager_opti, eager_loss
            planmod.clear_pack_cache()  # the weights changed behind torch's version counters during the replays
    if args.profile:
        print(f"epoch time: {trainTime * args.epochs:.4f}")
        return
    model.eval()
    sync()
    start = time.time()
    with torch.no_grad():
        for _ in range(args.epochs):
            Z = model(X)
    sync()
    inferenceTime = (time.time() - start) / args.epochs
    if graph_infer:
        # The forward of a small model is a chain of launch-bound kernels: capture it once (every plan, bound
        # scale set and packed weight exists after the eager passes above) and replay the graph per forward.
        eager = Z.clone()
        static_x = X.clone()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):
                model(static_x)
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g):
            static_z = model(static_x)
        g.replay()
        sync()
        assert torch.allclose(static_z, eager, rtol=1e-5, atol=1e-6), "graph replay differs from the eager forward"
        start = time.time()
        for _ in range(args.epochs):
            g.replay()
        sync()
        graphTime = (time.time() - start) / args.epochs
        if rank == 0:
            print(f"backend {args.backend}: avg inference time {graphTime:.6f} as a hipGraph replay (eager {inferenceTime:.6f})")
        inferenceTime = graphTime
    if rank == 0:
        assert torch.isfinite(loss), "training diverged"
        print(f"backend {args.backend}: avg epoch time {trainTime:.4f}")
        print(f"backend {args.backend}: avg inference time {inferenceTime:.6f} (final loss {loss.item():.4f}, "
              f"{world} rank(s), {inc.N} vertices / {inc.M} hyperedges / {inc.nnz} incidences per rank)")
        if args.output is not None:
            with open(args.output, "a") as f:
                print(f"{args.backend},{args.model},{args.dname},nlayer={args.nlayer}, nhid={args.nhid}, "
                      f"nhead={args.nhead},first_aggr={args.first_aggr},{trainTime},{inferenceTime}", file=f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
