"""bench.py keeps the driver's contract: one JSON line with the metric, the roofline of the dominant kernel and the CPU baseline at N = 1,
and the N-rank path (base rows sharded, one all-gather, merge) gives the single-engine result.  The N = 2 case shares the one GPU of
the test box between two ranks over gloo (NWKNN_DIST_BACKEND; production is nccl == RCCL, one rank per GPU)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _json_lines(out: str):
    return [json.loads(line) for line in out.splitlines() if line.startswith("{")]


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "dev", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    b = lines[0]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in b, key
    assert b["n_gpus"] == 1 and b["steps"] == 2 and b["warmup"] == 1 and b["unit"] == "queries/s" and b["higher_is_better"] is True
    assert b["vs_baseline"] is None and b["dtype"] == "f32" and "workload" in b["config"] and "model" not in b["config"]
    assert abs(b["value"] - 4096 / (b["ms_per_step"] / 1e3)) < 1e-6 * b["value"]
    rf = b["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3 and 0.0 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["launches"] > 0 and rf["avg_launch_ms"] > 0.0
    assert rf["kernel"] == "nwk_tile_kernel<0,false,256,16,0,3>"          # the kernel that ran, as rocprofv3 spells it
    cb = b["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "queries/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    sk = cb["sklearn"]
    assert sk["unit"] == "queries/s" and sk["value"] > 0 and sk["cores"] >= 1 and "NearestNeighbors" in sk["sample"]
    pre = b["exact_prescreen"]
    assert pre["identical_to_direct"] is True and pre["value"] > 0 and pre["roofline"]["peak"] == 2500.0
    assert pre["roofline"]["kernel"] == "nwk_tile_kernel_f16y<0,1>"
    assert b["ranks"] == 1 and len(b["devices"]) == 1 and b["devices"][0]["rank"] == 0 and "gfx950" in b["devices"][0]["gcn_arch"]
    assert "collective" not in b and "secondary" not in b                  # N = 1, not the default workload
    hbm = rf["hbm"]                                                         # north_star: achieved HBM GB/s on the streaming base read (the layout pass)
    assert hbm["bound"] == "hbm" and hbm["unit"] == "GB/s" and hbm["peak"] == 8000.0 and 0.0 < hbm["frac"] < 1.0 and hbm["kernel"] == "nwk_prep_tiled_kernel"
    inv = pre["invariant"]                                                  # every rescored pair held against the bound it was pruned under
    assert inv["violations"] == 0 and inv["fallbacks"] == 0 and inv["checked_pairs"] > 0 and 0.0 < inv["worst_error_over_bound"] < 1.0


@pytest.mark.gpu
def test_default_workload_carries_the_other_baseline_configs():
    """The default (C3) line at N = 1 ends with a time-boxed `secondary` block: the C4 per-GPU share (all 1M queries x one 12.5M-row shard),
    the reference's one published run (10k x 4M x 1536: in memory, and through compute_knn from a parquet file -- shrunk to 200k base
    rows here, the driver runs the 4M), C2 and one of eight C3 shards, each with its own value / ms_per_step / roofline (kernel,
    achieved, peak, frac).  One timed step here; the driver runs 20."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--files-base-rows", "200000"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    b = lines[0]
    assert b["metric"].startswith("queries/sec + pairwise-dist TFLOP/s, 100k x 10M") and b["config"]["base_rows"] == 10_000_000
    assert b["exact_prescreen"]["identical_to_direct"] is True
    sec = b["secondary"]
    assert sorted(sec) == ["c2", "c3shard", "c4gpu", "refcli"]
    for name, peak, kernel in (("c2", 157.3, "nwk_tile_kernel<"), ("c3shard", 157.3, "nwk_tile_kernel<"), ("c4gpu", 2500.0, "nwk_tile_kernel_f16"),
                               ("refcli", 157.3, "nwk_tile_kernel<")):
        s = sec[name]
        assert "error" not in s, s
        assert s["value"] > 0 and s["ms_per_step"] > 0 and s["unit"] == "queries/s" and s["steps"] >= 1
        rf = s["roofline"]
        assert rf["peak"] == peak and rf["kernel"].startswith(kernel) and 0.0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / peak) < 1e-9
    assert sec["c2"]["exact_prescreen"]["identical_to_direct"] is True and sec["c3shard"]["exact_prescreen"]["identical_to_direct"] is True
    assert sec["c4gpu"]["dtype"] == "f16" and sec["c2"]["config"]["base_rows"] == 1_000_000
    c4 = sec["c4gpu"]                                                       # one GPU's REAL share of config 4: every query, its shard
    assert c4["config"]["queries"] == 1_000_000 and c4["config"]["base_rows"] == 12_500_000 and c4["config"]["dim"] == 1536
    assert c4["invariant"]["violations"] == 0 and c4["invariant"]["fallbacks"] == 0 and c4["invariant"]["checked_pairs"] > 100_000_000
    assert c4["breakdown_ms_per_step"]["overflowed_queries"] == 0 and "50000 queries" in c4["warmup_note"]
    ref = sec["refcli"]
    assert ref["config"]["queries"] == 10_000 and ref["config"]["base_rows"] == 4_000_000 and ref["exact_prescreen"]["identical_to_direct"] is True
    pub = ref["vs_reference_published"]
    assert pub["reference"]["knn_stage_s"] == 103.46 and "cross-hardware" in pub["label"] and pub["ratio_in_memory"] > 1.0
    tf = ref["through_files"]
    assert "error" not in tf and tf["base_rows"] == 200_000 and tf["direct"]["knn_stage_s"] > 0 and tf["prescreened"]["knn_stage_s"] > 0
    assert tf["encoding"] == "uncompressed" and tf["direct_compute_knn_ds"]["knn_stage_s"] > 0 and tf["prescreened_compute_knn_ds"]["knn_stage_s"] > 0
    enc = tf["reference_encoding"]                                          # the same job from a file written as the reference's ParquetStreamer writes
    assert "error" not in enc and enc["encoding"] == "pyarrow_defaults" and enc["direct"]["knn_stage_s"] > 0 and "generate_dataset.py:241" in enc["encoding_note"]
    assert b["roofline"]["hbm"]["achieved"] > 1000.0                        # the layout pass over 41 GB of base: TB/s, not GB/s
    assert b["wall_s_since_process_start"] < 440.0 + 60.0                   # no leg starts after 440 s; the last one may run on


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["base", "queries"])
def test_bench_two_ranks_over_gloo_match_the_single_engine(shard):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, NWKNN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--workload", "dev", "--steps", "1", "--warmup", "1",
                        "--shard", shard, "--verify"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = _json_lines(r.stdout)
    bench = [x for x in lines if "metric" in x]
    verify = [x for x in lines if "verify" in x]
    assert len(bench) == 1 and bench[0]["n_gpus"] == 2 and bench[0]["scaling"] == "strong" and "cpu_baseline" not in bench[0]
    assert ("sharded x2" in bench[0]["config"]["parallelism"])
    assert len(verify) == 1 and verify[0]["ok"] is True


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_when_typed_as_the_driver_types_it():
    """`python bench.py --gpus 2 ...` with NO torchrun around it: the script starts its two ranks itself (a child process, before
    anything touches the GPU) and the line proves two ranks ran: `ranks`, one `devices` entry per rank, the collective's backend and
    timings.  gloo here (two ranks share the box's one GPU); nccl == RCCL is the default."""
    env = dict(os.environ, NWKNN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(var, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "dev", "--steps", "1", "--warmup", "1", "--verify"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = _json_lines(r.stdout)
    bench = [x for x in lines if "metric" in x]
    verify = [x for x in lines if "verify" in x]
    assert len(bench) == 1 and len(verify) == 1 and verify[0]["ok"] is True
    b = bench[0]
    assert b["n_gpus"] == 2 and b["ranks"] == 2 and [d["rank"] for d in b["devices"]] == [0, 1]
    assert all("gfx950" in d["gcn_arch"] for d in b["devices"])
    c = b["collective"]
    assert c["backend"] == "gloo" and c["allgather_ms"] > 0.0 and c["merge_ms"] > 0.0 and c["allgather_bytes_per_rank"] == 2 * 4096 * 100 * 12
    assert len(c["rank_kernel_ms_per_step"]) == 2 and all(x > 0.0 for x in c["rank_kernel_ms_per_step"])
    assert "torch.distributed" in c["kind"] and c["sharding"].startswith("base rows") and "query slice" in c["sharding"] and c["slices_allgather_ms"] > 0.0
    pre = b["exact_prescreen"]                                     # the pre-screened leg shards the QUERY rows (the whole base fits a GPU): same bits
    assert pre["identical_to_direct"] is True and pre["collective"]["sharding"].startswith("query rows") and pre["collective"]["merge_ms"] == 0.0
    assert "torch.distributed.run" in r.stderr                     # the parent said what it started


@pytest.mark.gpu
def test_bench_over_rccl_refuses_more_ranks_than_gpus():
    """The production backend is nccl == RCCL, one rank per GPU: asking for more ranks than the node has GPUs fails loudly in every
    rank (no silent device sharing), and the self-launching parent hands the failure through as its exit code."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "NWKNN_DIST_BACKEND")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--workload", "dev", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert f"needs {n} GPUs" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_bench_drives_the_in_library_collective():
    """--collective inlib: the bench line produced through the C ABI's own multi-GPU entry points (nwknn_comm_create / generate_base /
    nwknn_search_sharded: RCCL inside libnwknn.so, one process) instead of torch.distributed -- with the one device this box has, checked
    against a single-engine search."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--collective", "inlib", "--workload", "dev", "--steps", "2", "--warmup", "1",
                        "--verify"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    bench = [x for x in lines if "metric" in x]
    verify = [x for x in lines if "verify" in x]
    assert len(bench) == 1 and len(verify) == 1 and verify[0]["ok"] is True
    b = bench[0]
    assert b["n_gpus"] == 1 and b["ranks"] == 1 and b["value"] > 0 and 0.0 < b["roofline"]["frac"] < 1.0
    c = b["collective"]
    assert "in-library" in c["kind"] and c["backend"] == "rccl" and c["allgather_ms"] > 0.0 and c["merge_ms"] > 0.0 and c["search_ms"] > 0.0
    assert c["allgather_bytes_per_rank"] == 4096 * 100 * 12
