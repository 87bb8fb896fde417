"""bench.py's output contract (CPU): the final stdout line is ONE JSON object below 4 KB that keeps the keys
the driver reads, whatever the run produced; `--gpus N` outside torchrun launches torchrun as a child process,
relays rank 0's line and the exit code; an exchange of the sharded section that raises fails the run.
The round-3 line was 20 KB, past the driver's 8 KB window: BENCH_r03.parsed was null."""
import copy
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports neither torch nor the package at module level)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "configs")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes_per_step", "avg_step_us")
CPU = ("value", "unit", "cores", "kind", "sample", "all_cores", "host")
RECORD = ("workload", "ms_per_step", "frac", "traffic_over_algorithmic", "parity_ok")


def _entry(shape, reps, F, weighted=False, lin=0, ms=0.15):
    balg = bench.b_alg(2708 * reps, 1579 * reps, 4859 * reps, F, 2 if weighted else 0, weighted)
    r = {"workload": bench.workload_name(shape, reps, F) + (", weighted (degE, degV, W)" if weighted else ""),
         "short": bench.short_name(shape, reps, F, weighted, lin),
         "op": "H*H^T*X (aggr_proto)", "feat_len": F, "variant": "auto", "resolved_variant": "fused",
         "ms_per_step": ms * 1.01, "device_ms_per_step": ms, "edges_per_s": 3.2e10,
         "roofline": {"bound": "hbm", "achieved": balg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                      "frac": balg / (ms * 1e-3) / 1e9 / 8000.0, "traffic": 0.94 * balg, "kernel": "fused_packed_kernel",
                      "algorithmic_bytes_per_step": balg, "avg_step_us": ms * 1e3, "launches_per_step": 1,
                      "helper_launches_per_step": 2,
                      "floor": {"note": "x" * 230, "ms_at_6300_GBs": 0.1}},
         "fused_schedule": {"panels": 123456, "n_mat": 7, "note": "y" * 500},
         "parity": {"ok": True, "bit_exact": False, "max_rel_err": 1.234567e-6, "max_rel_err_vs_float64": 5.9e-7,
                    "bound": "b" * 100, "against": "a" * 70, "float64_ok": True}}
    if lin:
        r["roofline_mfma"] = {"bound": "mfma", "achieved": 64.4, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.41}
    return r


def canned_result(n_gpus=1, sharded=False, broken=False):
    """A full result as bench.main() assembles it, every optional section present, strings at their longest."""
    head = _entry("cora", 1024, 32)
    cells = [("citeseer", 1024, 32, False, 0), ("pubmed", 256, 32, False, 0), ("cora", 256, 128, False, 0),
             ("citeseer", 256, 128, False, 0), ("pubmed", 64, 128, False, 0), ("pubmed", 64, 128, False, 128),
             ("powerlaw", 1, 64, False, 0), ("cora", 1024, 32, True, 0)]
    configs = [copy.deepcopy(head)] + [_entry(*c) for c in cells]
    if broken:
        configs[3] = {"workload": bench.workload_name("cora", 256, 128), "short": bench.short_name("cora", 256, 128),
                      "error": "RuntimeError: " + "z" * 300}
    out = {"metric": "aggregated edges/sec (fused V->E->V aggregation)", "value": 3.257e10 * n_gpus, "unit": "edges/s",
           "n_gpus": n_gpus, "steps": 200, "warmup": 20, "ms_per_step": 0.1528, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": head["workload"], "op": head["op"], "vertices_per_gpu": 2772992, "hyperedges_per_gpu": 1616896,
                      "nnz_per_gpu": 4975616, "feat_len": 32, "variant": "auto", "resolved_variant": "fused",
                      "sharding": "hyperedge groups (one hypergraph batch per rank), no data-path collective"},
           "roofline": head["roofline"], "hbm_gbs_algorithmic": 5022.1, "plan_build_s": 0.04, "fused_schedule_build_s": 0.6,
           "fused_schedule": head["fused_schedule"], "plan": {"panels": 1}, "detail_path": "/somewhere/bench_detail.json",
           "parity": head["parity"], "configs_detail": configs,
           "device_copy": {"ms": 0.133, "gbs": 5321.0, "step_over_copy": 1.15, "note": "n" * 100},
           "single_graph": {"fused_us": 3.8, "pull_us": 11.0, "note": "m" * 100},
           "comparator": {name: {"native_us": 3.8, "rocsparse_twostep_us": 6.0, "speedup_over_rocsparse_twostep": 1.58,
                                 "reference_fused_us_rtx3090": 4.79, "reference_cusparse_twostep_us_rtx3090": 40.7,
                                 "reference_speedup_over_cusparse_twostep": 8.48} for name in ("cora", "citeseer", "pubmed")},
           "cpu_baseline": {"value": 1.217e7, "unit": "edges/s", "cores": 1, "kind": "port",
                            "sample": "first 1024 of 1024 cora-shape hypergraphs of the batch, F=32, best of 24 passes (0.409 s each)",
                            "all_cores": {"value": 1.147e8, "cores": 128}, "host": "256 logical cpus"}}
    out["comparator"]["note"] = "c" * 200
    if sharded:
        form = {"ms_per_step": 3.21, "value": 5.9e9, "unit": "edges/s", "local_kernel_ms": 0.31, "shard_hyperedges": [0, 500000],
                "max_rel_err_vs_single_gpu": 6.9e-7, "ok": True}
        out["sharded"] = {"workload": "w" * 80, "vertices": 1000000, "hyperedges": 4000000, "feat_len": 64, "nnz": 19100000,
                          "scaling": "strong", "rccl_ranks": n_gpus, "backend": "nccl", "errors": int(broken),
                          "single_gpu_ms": 0.91, "single_gpu_value": 2.1e10,
                          "allreduce": dict(form), "reduce_scatter": dict(form),
                          "allreduce_pipelined": dict(form, column_chunks=2),
                          "column_sharded": {"error": "RuntimeError: " + "q" * 300} if broken else dict(form, columns_per_rank=8)}
    return out


@pytest.mark.parametrize("n_gpus,sharded,broken", [(1, False, False), (1, True, False), (8, True, False), (8, True, True),
                                                   (1, False, True)])
def test_final_line_fits_and_keeps_the_keys(n_gpus, sharded, broken):
    out = canned_result(n_gpus, sharded, broken)
    line = bench.compact_line(out)
    assert "\n" not in line
    assert len(line.encode()) < 4096, len(line)
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    for k in ROOFLINE:
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - out["roofline"]["frac"]) < 1e-3
    for k in CPU:
        assert k in d["cpu_baseline"], k
    assert "informative" in d["cpu_baseline"]["all_cores"]["note"]
    assert set(d["parity"]) >= {"ok", "max_rel_err", "max_rel_err_vs_float64"}
    assert "workload" in d["config"]
    assert len(d["configs"]) == 9  # the six target cells (headline included), the MFMA path, config 4, the weighted operator
    for rec in d["configs"]:
        if "error" in rec:
            assert rec["parity_ok"] is False
            continue
        for k in RECORD:
            assert k in rec, (rec, k)
    assert [r for r in d["configs"] if "mfma_frac" in r], "the aggregation + linear entry carries its MFMA fraction"
    assert d["parity"]["all_configs_ok"] == (not broken)
    assert set(d["single_graph"]) >= {"cora", "citeseer", "pubmed"}
    assert d["single_graph"]["cora"]["speedup_over_rocsparse_twostep"] == pytest.approx(1.58)
    assert d["single_graph"]["cora"]["reference_speedup_over_cusparse"] == pytest.approx(8.48)
    if sharded:
        ss = d["strong_scaling"]
        assert ss["rccl_ranks"] == n_gpus and ss["backend"] == "nccl"
        for form in ("allreduce", "reduce_scatter", "allreduce_pipelined", "column_sharded"):
            assert form in ss
            if "error" not in ss[form]:
                assert "max_rel_err_vs_single_gpu" in ss[form] and "value" in ss[form]
        assert ss["errors"] == int(broken)


def test_round3_result_fits():
    """The result that overflowed the driver's window in round 3 (profiles/r03_bench_default.json, 20 251 bytes as printed
    then), through this round's line builder."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")))
    head = dict(d, device_ms_per_step=d["ms_per_step"], workload=d["config"]["workload"])
    d["configs_detail"] = [head] + d.pop("configs")
    line = bench.compact_line(d)
    assert len(line.encode()) < 4096
    got = json.loads(line)
    assert len(got["configs"]) == 9 and got["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-3)


def _run_bench(*flags):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env,
                       timeout=300)
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    return r, lines


def test_gpus_2_self_launches_torchrun_and_relays_the_line(tmp_path):
    """`python bench.py --gpus 2`, invoked plainly: a child `python -m torch.distributed.run --nproc-per-node 2`, two ranks
    that meet over gloo (--rehearse-cpu: control flow only, no GPU, no kernels), max over ranks, rank 0's line last."""
    detail = str(tmp_path / "detail.json")
    r, lines = _run_bench("--gpus", "2", "--rehearse-cpu", "--steps", "5", "--warmup", "0", "--detail", detail)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "torch.distributed.run" in r.stderr
    d = json.loads(lines[-1])
    assert len(lines[-1].encode()) < 4096
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["data"].startswith("rehearsal")
    # the slower rank (it sleeps twice as long per step) sets the time: max over ranks
    assert d["ms_per_step"] >= 0.9
    ss = d["strong_scaling"]
    assert ss["rccl_ranks"] == 2 and ss["backend"] == "gloo" and ss["errors"] == 0
    assert ss["allreduce"]["max_rel_err_vs_single_gpu"] == 0.0
    full = json.load(open(detail))
    assert full["sharded"]["allreduce"]["ok"] and "configs_detail" in full


def test_a_failing_exchange_fails_the_run(tmp_path):
    r, lines = _run_bench("--gpus", "2", "--rehearse-cpu", "--steps", "3", "--warmup", "0", "--inject-exchange-error",
                          "--detail", str(tmp_path / "detail.json"))
    assert r.returncode != 0
    d = json.loads(lines[-1])  # the line is still printed, with the error on it
    assert "error" in d["strong_scaling"]["reduce_scatter"] and d["strong_scaling"]["errors"] >= 1


def test_gpus_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-cpu"], capture_output=True,
                       text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
