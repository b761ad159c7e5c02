#!/usr/bin/env python3
"""bench.py -- headline benchmark of the brute-force kNN hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|...]

With N > 1 and no RANK in the environment the script starts its own N ranks (a child `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...`, one rank per GPU over RCCL) before
anything touches the GPU, relays the child's output and exits with its code; launched under torch.distributed.run by the caller
it joins the existing rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

Metric (BASELINE.json): queries/s (+ pairwise-distance TFLOP/s) for 100k queries x 10M base rows, d=1024, k=100,
fp32 squared-L2 ("c3"; 41 GB of base fits one 288 GB MI355X, so N=1 runs the same job).  One *step* is one
complete ground-truth job: raw row-major float32 queries + base resident in HBM -> layout/norm prep -> distance
tiles fused with the running top-k -> (N>1: one RCCL all-gather of the [Q,k] partial lists + merge) -> final
[Q,k] distances+indices in host memory on rank 0.  With N ranks the base rows are sharded N ways (total work
fixed: strong scaling); queries are replicated.  Inputs are seeded synthetic embeddings generated on device.

The JSON line also carries
  roofline     -- distance-tile kernel: algorithmic FLOPs (2*Q*B*d per launch set) / its summed HIP-event time,
                  against the dense f32 / f16 MFMA peak of MI355X_MICROARCH.md; `kernel` is the kernel that ran, as rocprofv3 spells it;
  cpu_baseline -- numpy/BLAS brute force (oracle.cpu_oracle.knn_blas_fp32) on a bounded sample, rank 0, N=1 only, with
                  sklearn.neighbors.NearestNeighbors(algorithm='brute') on a second sample beside it (`cpu_baseline.sklearn`);
  ranks / devices / collective -- what torch.distributed saw: world size, every rank's GPU, backend and the per-step time of the
                  all-gather and of the merge (N > 1);
  secondary    -- N=1, default workload only: the other BASELINE configs one GPU can hold (the C4 per-GPU share: all 1M queries x one
                  12.5M-row shard; the reference's one published run, 10k x 4M x 1536, in memory and through compute_knn from a parquet
                  file; C2; one of eight C3 shards), a few steps each with their own value / ms_per_step / roofline, on the same clock as
                  the headline.  The block is time-boxed (SECONDARY_BUDGET_S of wall clock since process start): a leg that would not fit
                  is reported as skipped, never silently dropped.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
T_PROCESS_START = time.perf_counter()

WORKLOADS = {
    # name: (Q, B, d, k, metric, dtype, description)
    "c3": (100_000, 10_000_000, 1024, 100, "sqeuclidean", "f32", "C3: 100k queries x 10M base, d=1024 (e5-large-v2 shape), k=100, fp32 sqeuclidean"),
    "c2": (10_000, 1_000_000, 768, 100, "sqeuclidean", "f32", "C2: 10k queries x 1M base, d=768, k=100, fp32 sqeuclidean"),
    "dev": (4_096, 262_144, 256, 100, "sqeuclidean", "f32", "dev: 4k x 256k, d=256, k=100 (not a headline number)"),
    # what ONE of eight GPUs does per step of the headline job at --gpus 8 (all queries x its 1.25M-row shard; no collective):
    # 8 x its queries/s is the ceiling of the 8-GPU line, measurable on a one-GPU box
    "c3shard": (100_000, 1_250_000, 1024, 100, "sqeuclidean", "f32", "one of eight C3 shards: 100k queries x 1.25M base rows, d=1024, k=100, fp32 (not a headline number)"),
    # ... and what one of eight GPUs does when the QUERY rows are sharded instead (how the pre-screened leg runs at --gpus 8: every GPU holds the
    # whole base and an eighth of the queries; no merge)
    "c3qshard": (12_500, 10_000_000, 1024, 100, "sqeuclidean", "f32", "one of eight C3 query shards: 12.5k queries x 10M base rows, d=1024, k=100, fp32 (not a headline number)"),
    # BASELINE config 4 (fp16 MFMA + L2, 8 GPUs) and the piece of it one GPU sees (its 12.5M-row shard, a tenth of the queries)
    "c4": (1_000_000, 100_000_000, 1536, 100, "sqeuclidean", "f16", "C4: 1M queries x 100M base, d=1536 (ada-002 shape), k=100, fp16 MFMA sqeuclidean"),
    # under north_star's base sharding every GPU sees ALL the queries against its shard
    "c4gpu": (1_000_000, 12_500_000, 1536, 100, "sqeuclidean", "f16", "C4 per-GPU share: all 1M queries x 12.5M base rows (one of 8 shards), d=1536, k=100, fp16 (exact fp16 mode)"),
    "c4slice": (100_000, 12_500_000, 1536, 100, "sqeuclidean", "f16", "a tenth of the C4 per-GPU share: 100k queries x 12.5M base rows, d=1536, k=100, fp16 (rounds 1-3 quoted this; not a headline number)"),
    # the one run the reference publishes (docs/cli.png): 10k x 4M, d=1536, k=100, fp32 sqeuclidean, RTX A5500 laptop GPU
    "refcli": (10_000, 4_000_000, 1536, 100, "sqeuclidean", "f32", "the reference's published run (docs/cli.png): 10k queries x 4M base, d=1536, k=100, fp32 sqeuclidean"),
    # BASELINE config 5: MaxSim late interaction (an extension -- the reference's `ck` is flat token kNN): Q = queries of 32 tokens,
    # B = docs of 64 tokens, d = 128, k = 1000; FLOPs = 2 * (32 Q) * (64 B) * d
    "c5": (2_048, 1_000_000, 128, 1000, "maxsim", "f32", "C5: MaxSim, 2048 queries x 32 tokens vs 1M docs x 64 tokens, d=128, k=1000, fp32"),
    "dev5": (256, 100_000, 128, 100, "maxsim", "f32", "dev5: MaxSim 256 queries vs 100k docs (not a headline number)"),
    "dev16": (32_768, 1_048_576, 1536, 100, "sqeuclidean", "f16", "dev16: 32k x 1M, d=1536, k=100, fp16 (not a headline number)"),
}
# (workload, timed steps, timed steps of its pre-screened leg, queries of the one warm-up step (0: all), estimated wall seconds)
SECONDARY = (("c4gpu", 1, 0, 50_000, 36), ("refcli", 2, 2, 0, 5), ("c2", 10, 2, 0, 3), ("c3shard", 3, 2, 0, 10))
SECONDARY_BUDGET_S = 440.0          # wall clock since process start after which no further secondary leg is started (the driver allows 600 s)
REF_PUBLISHED = {"knn_stage_s": 103.46, "queries_per_s": 96.7, "tflops": 1.19, "hardware": "1x NVIDIA RTX A5500 Laptop GPU 16 GB",
                 "source": "docs/cli.png (BASELINE.md section 1): kNN stage incl. parquet load, 9 x 9 knn() calls"}
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E ~8 TB/s
SEED_BASE, SEED_QUERY = 0x6E770001, 0x6E770002
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0}      # MI355X_MICROARCH.md: dense f32 / f16 MFMA peaks


def progress(msg: str) -> None:
    """One line on stderr (stdout carries the JSON line only): a harness that takes a silent process for a hung one sees the run move."""
    if os.environ.get("RANK", "0") == "0":
        print(f"bench.py [{time.perf_counter() - T_PROCESS_START:6.0f} s] {msg}", file=sys.stderr, flush=True)


def profiler_preload() -> str:
    """Non-empty when this process runs under rocprofv3 / rocprofiler-sdk: the tool's preloaded library initialises the GPU before main(),
    so starting the ranks from here would be an exec from a process that has touched the GPU (forbidden on this pool: it takes the
    machine down).  Returns what gave it away."""
    for var in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY", "ROCP_TOOL_LIB", "HSA_TOOLS_LIB"):
        val = os.environ.get(var, "")
        if any(t in val.lower() for t in ("rocprof", "roctracer", "rocprofiler")):
            return f"{var}={val}"
    for var in os.environ:
        if var.startswith(("ROCPROF_", "ROCPROFILER_", "ROCPROFV3_")):
            return f"{var} is set"
    return ""


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` as typed: start the N ranks as a CHILD process (never an exec: this process has not touched the
    GPU and never will), pass its output through, return its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the pool's host driver supports dmabuf IPC only (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    print(f"bench.py: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def recorded_traffic(workload):
    """(HBM/fabric bytes per launch of the tile kernel, which kernel it was measured on) from the committed rocprofv3 PMC passes (FETCH_SIZE
    doubled per the gfx950 correction + WRITE_SIZE, separate passes; profiles/traffic_<workload>.json, written by tools/profile_round.sh).
    PMC cannot be sampled from inside this process, so the figure is the recorded one for this workload, or (None, None)."""
    f = ROOT / "profiles" / f"traffic_{workload}.json"
    try:
        rec = json.loads(f.read_text())
        tile = rec.get("tile", rec)
        return float(tile["hbm_fabric_bytes_per_launch"]), str(tile.get("kernel", ""))
    except Exception:
        return None, None


def _blas_threads():
    """(BLAS threads the CPU baseline will use, a threadpoolctl controller or None).  The GPU box shows 256 CPUs and allows 16 CPUs' worth of time
    (cgroup quota): a BLAS pool sized by the former spends the quota on contention, so the pool is limited to what the container may run --
    `cores` in the line is that number.  The controller is made HERE, on the main thread (it walks the loaded libraries under the loader lock)."""
    from neighborhoodwatch_amd.hostinfo import effective_cpus
    cores = effective_cpus()
    try:
        from threadpoolctl import ThreadpoolController
        ctl = ThreadpoolController()
        have = max([p.get("num_threads", 1) for p in ctl.info() if p.get("user_api") == "blas"] or [cores])
        return min(have, cores), ctl
    except Exception:
        return cores, None


def cpu_baseline(base_dev, queries_dev, k, metric, full_b, flop_budget=0.45e13, defer=False, blas_threads=None):
    """Brute force on the host cores over a bounded sample of the same workload: numpy/BLAS (the oracle's sgemm form) and, beside it,
    scikit-learn's NearestNeighbors(algorithm='brute').  Each leg is sized to ~flop_budget of sgemm (~10 s on a 64-core host).
    defer=True (the default line, which has GPU-only legs still to run): the sample is copied to the host now, the ~25 s of host work run
    on a background thread, and a concurrent.futures.Future of the result is returned (main() resolves it before it prints)."""
    if defer:
        # Everything that imports a module or walks the loaded libraries happens HERE, on the main thread: threadpoolctl's threadpool_info()
        # runs Python inside dl_iterate_phdr (it holds the loader lock and needs the GIL) while another thread's first `import pyarrow` holds
        # the GIL and needs the loader lock for its dlopen -- the default line deadlocked on exactly that the first time this was deferred.
        from concurrent.futures import ThreadPoolExecutor
        import numpy  # noqa: F401
        from oracle import cpu_oracle  # noqa: F401
        try:
            from sklearn.neighbors import NearestNeighbors  # noqa: F401
        except Exception:
            pass
        blas_threads = _blas_threads()
        d_ = base_dev.shape[1]
        qs_ = min(queries_dev.shape[0], 2048)
        bs_ = min(base_dev.shape[0], max(10_000, int(flop_budget / (2.0 * qs_ * d_))))
        b_host, q_host = base_dev[:bs_].float().cpu(), queries_dev[:qs_].float().cpu()
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="nw-bench-cpu")
        fut = pool.submit(cpu_baseline, b_host, q_host, k, metric, full_b, flop_budget, False, blas_threads)
        pool.shutdown(wait=False)
        return fut
    import contextlib
    import numpy as np
    from oracle import cpu_oracle
    if blas_threads is None:
        blas_threads = _blas_threads()
    cores, ctl = blas_threads
    limited = ctl.limit(limits=cores, user_api="blas") if ctl is not None else contextlib.nullcontext()
    with limited:
        return _cpu_baseline_legs(base_dev, queries_dev, k, metric, full_b, flop_budget, cores)


def _cpu_baseline_legs(base_dev, queries_dev, k, metric, full_b, flop_budget, cores):
    import numpy as np
    from oracle import cpu_oracle
    d = base_dev.shape[1]
    qs = min(queries_dev.shape[0], 2048)
    bs = min(base_dev.shape[0], max(10_000, int(flop_budget / (2.0 * qs * d))))
    base = base_dev[:bs].float().cpu().numpy()
    qry = queries_dev[:qs].float().cpu().numpy()
    cpu_oracle.knn_blas_fp32(base[: min(bs, 20000)], qry[:64], min(k, 100), metric)      # warm-up
    t0 = time.perf_counter()
    cpu_oracle.knn_blas_fp32(base, qry, k, metric)
    dt = time.perf_counter() - t0
    scale = bs / float(full_b)                                                   # brute force is linear in B
    out = {
        "value": qs / dt * scale, "unit": "queries/s", "cores": int(cores), "kind": "port",
        "cores_note": f"BLAS threads = the CPUs this container may use (cgroup quota / affinity); the host shows {os.cpu_count()}",
        "sample": f"{qs} queries x first {bs} of {full_b} base rows in {dt:.2f}s (numpy sgemm + argpartition, "
                  f"oracle.cpu_oracle.knn_blas_fp32); rate scaled by {bs}/{full_b} to the full base",
    }
    if metric in ("sqeuclidean", "cosine"):
        try:
            from sklearn.neighbors import NearestNeighbors
            nn = NearestNeighbors(n_neighbors=min(k, bs), algorithm="brute", metric=metric, n_jobs=cores)
            nn.fit(base[: min(bs, 20000)]).kneighbors(qry[:64], return_distance=True)     # warm-up
            t0 = time.perf_counter()
            nn.fit(base).kneighbors(qry, return_distance=True)
            dts = time.perf_counter() - t0
            out["sklearn"] = {"value": qs / dts * scale, "unit": "queries/s", "cores": cores, "kind": "library",
                              "sample": f"sklearn.neighbors.NearestNeighbors(algorithm='brute', metric='{metric}', n_jobs={cores}).kneighbors on the same "
                                        f"{qs} x {bs} sample in {dts:.2f}s; same scaling"}
        except Exception as e:                                                    # the baseline is a report, never a reason to lose the line
            out["sklearn"] = {"error": f"{type(e).__name__}: {e}"}
    return out


REF_FILES_EST_S = 30               # the file legs of the reference's published run: write a 24.6 GB parquet file, run compute_knn / compute_knn_ds from it; then a 3 GB file in pyarrow's default encoding
REF_FILES_BASE_ROWS = 4_000_000


def ref_published_block(value_qps, pre):
    """The reference's one published figure for this path next to ours -- DIFFERENT hardware (a 16 GB laptop GPU), quoted as context."""
    out = {"reference": REF_PUBLISHED, "ours_queries_per_s_in_memory": value_qps, "ratio_in_memory": value_qps / REF_PUBLISHED["queries_per_s"],
           "label": "cross-hardware; the reference's figure includes its parquet load (see through_files for ours with load)"}
    if pre:
        out["ours_prescreened_queries_per_s"] = pre["value"]
    return out


FILE_ENCODINGS = {
    # bench.py's first file leg since round 2: the cheapest bytes to decode
    "uncompressed": dict(compression="NONE", use_dictionary=False, write_statistics=False),
    # what the reference's own writer produces: ParquetStreamer opens pq.ParquetWriter(filename, schema) with pyarrow's defaults
    # (generate_dataset.py:241-242,255-256), i.e. snappy + dictionary pages + statistics; synth.py writes the same way
    "pyarrow_defaults": {},
}
REF_FILES_DEFAULTS_ROWS = 500_000  # rows of the default-encoding leg: pyarrow writes snappy + dictionary at ~0.15 GB/s on one thread (no writer threads in
                                   # pyarrow 25), so the 4M-row file would take ~170 s to WRITE (not timed, but inside the bench's wall clock)


def synth_column_chunks(eng, seed, rows, d):
    """[d, n] float32 host arrays, one per 100 000-row group: column j of a group is one contiguous array (what a parquet column chunk
    is written from).  Generated on the GPU -- 614 MB per group of the reference's shape, ~25 ms of generate + transpose + copy."""
    return [eng.generate(seed, r0, min(100_000, rows - r0), d).t().contiguous().cpu().numpy() for r0 in range(0, rows, 100_000)]


def write_ref_files(eng, base_rows, q_rows=10_000, d=1536, encoding="uncompressed", chunks=None):
    """The synthetic source files of a file leg (one float32 column per dimension, 100 000-row groups, /dev/shm when it is there).
    `chunks` = (base groups, query groups) from synth_column_chunks when the values were generated ahead: the writing itself is host work
    only, and the default line does it on a background thread under the c4gpu leg."""
    import tempfile
    import pyarrow as pa
    import pyarrow.parquet as pq
    root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="nwk_refcli_", dir=root)
    out = {"data_dir": tmp, "base_rows": base_rows, "queries": q_rows, "dim": d, "encoding": encoding}
    t0 = time.perf_counter()
    names = [f"embedding_{j}" for j in range(d)]
    try:
        if chunks is None:
            chunks = (synth_column_chunks(eng, SEED_BASE, base_rows, d), synth_column_chunks(eng, SEED_QUERY, q_rows, d))
        for fname, groups in (("base.parquet", chunks[0]), ("query.parquet", chunks[1])):
            writer = None
            for cols in groups:
                t = pa.Table.from_arrays([pa.array(cols[j]) for j in range(d)], names=names)
                if writer is None:
                    writer = pq.ParquetWriter(os.path.join(tmp, fname), t.schema, **FILE_ENCODINGS[encoding])
                writer.write_table(t)
            writer.close()
        out["write_s_not_timed"] = time.perf_counter() - t0
        out["file_gb"] = os.path.getsize(os.path.join(tmp, "base.parquet")) / 1e9
        out["float_gb"] = base_rows * d * 4 / 1e9
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def ref_cli_through_files(eng, base_rows=REF_FILES_BASE_ROWS, q_rows=10_000, d=1536, k=100, encoding="uncompressed", legs=("direct", "prescreened"),
                          via=("compute_knn",), prepared=None):
    """The reference's published run as its CLI does it: compute_knn(data_dir, ...) from parquet files with one float32 column per
    dimension (cu_knn.py:151-180), load included, then merge_indices_and_distances -- what docs/cli.png times at 103.46 s + 2.02 s.
    The synthetic source files are written first (not timed: the reference's stage 1-2 is not in its figure either; `prepared` = the result
    of write_ref_files when a background thread already wrote them), in the named ENCODING (FILE_ENCODINGS: uncompressed, or pyarrow's
    defaults as the reference's ParquetStreamer writes).  `via` adds compute_knn_ds (the dataset flavour, cu_knn_ds.py:136-146)."""
    import shutil
    from neighborhoodwatch_amd import cu_knn, cu_knn_ds, merge
    out = prepared if prepared is not None else write_ref_files(eng, base_rows, q_rows, d, encoding)
    tmp = out["data_dir"]
    out["k"] = k
    out["encoding_note"] = ("pyarrow defaults (snappy, dictionary, statistics): what the reference's ParquetStreamer writes, generate_dataset.py:241-242"
                            if out["encoding"] == "pyarrow_defaults" else
                            "compression NONE, no dictionary, no statistics: the cheapest bytes to decode, NOT what the reference writes")
    try:
        if "error" in out:
            return out
        base_rows, q_rows, d = out["base_rows"], out["queries"], out["dim"]
        for how in via:
            for label in legs:
                pre = label == "prescreened"
                shutil.rmtree(os.path.join(tmp, "partial"), ignore_errors=True)
                os.makedirs(os.path.join(tmp, "partial"))                    # the CLI creates it (nw_utils.py:67-73)
                with contextlib.redirect_stdout(sys.stderr):                 # the drivers print progress like the reference; stdout carries the JSON line only
                    t0 = time.perf_counter()
                    if how == "compute_knn":
                        cu_knn.compute_knn(tmp, "synthetic", d, "query.parquet", q_rows, "base.parquet", base_rows, mem_tune=True, k=k, prescreen=pre)
                    else:
                        cu_knn_ds.compute_knn_ds(tmp, d, "query.parquet", q_rows, "base.parquet", base_rows, mem_tune=True, k=k, prescreen=pre)
                    t1 = time.perf_counter()
                    merge.merge_indices_and_distances(tmp, k)
                    t2 = time.perf_counter()
                key = label if how == "compute_knn" else f"{label}_compute_knn_ds"
                out[key] = {"knn_stage_s": t1 - t0, "merge_s": t2 - t1, "queries_per_s": q_rows / (t1 - t0),
                            "pairwise_tflops": 2.0 * q_rows * base_rows * d / (t1 - t0) / 1e12,
                            "decode_upload_gb_per_s": out["float_gb"] / (t1 - t0)}
                if base_rows == REF_FILES_BASE_ROWS:
                    out[key]["vs_reference_knn_stage"] = REF_PUBLISHED["knn_stage_s"] / (t1 - t0)
        out["label"] = ("cross-hardware (reference: RTX A5500 laptop GPU) AND, unless encoding says pyarrow_defaults, cross-encoding; parquet decode + PCIe "
                        "upload + search + partial files, as the reference's figure; row groups are decoded by reader threads and handed to the GPU as they "
                        "arrive (nwknn_base_begin / _append_column_chunks / _end)")
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


class RefFiles:
    """Both file legs of the refcli workload -- the reference's size on uncompressed files (compute_knn and compute_knn_ds) and a smaller
    file in the encoding the reference's own writer produces.  The VALUES are generated here, on the calling thread, while the GPU is
    free (27.7 GB of float32 through PCIe: a few seconds); the 29 GB of parquet are then written by a BACKGROUND thread -- host work only --
    while the caller goes on measuring (the default line: under the c4gpu leg, whose ~30 s of GPU work leave the host idle)."""

    def __init__(self, eng, base_rows, q_rows=10_000, d=1536):
        import threading
        import pyarrow  # noqa: F401  (first imports belong on the main thread: see cpu_baseline)
        import pyarrow.parquet  # noqa: F401
        from neighborhoodwatch_amd import cu_knn, cu_knn_ds, merge  # noqa: F401
        self.base_rows = base_rows
        self.small = min(base_rows, REF_FILES_DEFAULTS_ROWS)
        self.files = {}
        t0 = time.perf_counter()
        base = synth_column_chunks(eng, SEED_BASE, base_rows, d)
        qry = synth_column_chunks(eng, SEED_QUERY, q_rows, d)
        n_small = (self.small + 99_999) // 100_000
        small = [g if (i + 1) * 100_000 <= self.small else g[:, :self.small - i * 100_000] for i, g in enumerate(base[:n_small])]
        self.generate_s = time.perf_counter() - t0

        def work():
            self.files["plain"] = write_ref_files(None, base_rows, q_rows, d, chunks=(base, qry))
            self.files["defaults"] = write_ref_files(None, self.small, q_rows, d, encoding="pyarrow_defaults", chunks=(small, qry))
        self.thread = threading.Thread(target=work, name="nw-bench-files", daemon=True)
        self.thread.start()

    def run(self, eng):
        t0 = time.perf_counter()
        self.thread.join()
        waited = time.perf_counter() - t0
        out = ref_cli_through_files(eng, via=("compute_knn", "compute_knn_ds"), prepared=self.files["plain"])
        out["reference_encoding"] = ref_cli_through_files(eng, legs=("direct",), prepared=self.files["defaults"])
        out["files_written"] = (f"values generated on the GPU in {self.generate_s:.1f} s, files written on a background thread while the previous leg "
                                f"ran; the main thread waited {waited:.1f} s for it")
        return out

    def discard(self):
        import shutil
        self.thread.join()
        for f in self.files.values():
            shutil.rmtree(f.get("data_dir", ""), ignore_errors=True)


def ref_cli_file_legs(eng, base_rows):
    return RefFiles(eng, base_rows).run(eng)


class Bench:
    def __init__(self, args):
        # before anything initialises the HIP runtime: the host driver of this pool only supports dmabuf IPC (RCCL needs it)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as td
        from neighborhoodwatch_amd import KnnEngine
        self.torch, self.td, self.args = torch, td, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                             "(or plain `python bench.py --gpus N`, which starts the ranks itself)")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
        # NWKNN_DIST_BACKEND=gloo rehearses the N-rank path on a box with fewer GPUs than ranks (ranks share devices, the
        # all-gather goes through the host); production is nccl == RCCL, one rank per GPU.
        self.backend = os.environ.get("NWKNN_DIST_BACKEND", "nccl") if self.world > 1 else "none"
        n_dev = torch.cuda.device_count()
        if self.backend == "nccl" and n_dev < self.world:
            raise SystemExit(f"--gpus {self.world} over RCCL needs {self.world} GPUs, this node shows {n_dev} "
                             "(NWKNN_DIST_BACKEND=gloo rehearses the N-rank path on fewer)")
        self.dev_index = local_rank if self.backend == "nccl" else local_rank % n_dev
        torch.cuda.set_device(self.dev_index)
        if self.world > 1:
            if self.backend == "nccl":
                td.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.dev_index))
            else:
                td.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        self.red_dev = "cuda" if self.backend == "nccl" else "cpu"
        self.eng = KnnEngine(self.dev_index, profile=True)
        for kvs in args.opt:
            for kv in kvs.split(","):
                name, value = kv.split("=")
                self.eng.set_option(name, int(value))

    # ---- what this device's matrix pipes deliver with nothing in the way ----------------------------------------------------------------
    def ceiling(self, dtype):
        """nwknn_mfma_ceiling, once per dtype and process: TFLOP/s and in-kernel clock of a bare MFMA loop on random operands (1 s of
        back-to-back launches, the second half timed) -- the power-limited rate of THIS device, which the nominal peak ignores."""
        if not hasattr(self, "_ceil"):
            self._ceil = {}
        if dtype not in self._ceil:
            try:
                self._ceil[dtype] = self.eng.mfma_ceiling(dtype, 1.0)
            except Exception as e:                                      # a calibration figure must never cost the line
                self._ceil[dtype] = {"error": f"{type(e).__name__}: {e}"}
        return self._ceil[dtype]

    def fabric(self):
        """nwknn_fabric_probe, once per process: GB/s of a plain streaming read over 128 MiB (beyond L2, inside the Infinity Cache: where the tile
        kernels' L2 misses are served) and over 4 GiB (HBM).  Round 4's driver box ran the fp32 tile kernel 2 % slower than the builder's boxes
        at the SAME bare MFMA rate; this is the figure that says whether a box's fabric is what differs."""
        if not hasattr(self, "_fabric"):
            try:
                self._fabric = {"fabric_probe_gbs": self.eng.fabric_probe(128 << 20, 0.05), "hbm_probe_gbs": self.eng.fabric_probe(4 << 30, 0.05),
                                "fabric_probe_note": "plain streaming read kernel, 50 ms each: 128 MiB working set (Infinity Cache) / 4 GiB (HBM)"}
            except Exception as e:
                self._fabric = {"fabric_probe_error": f"{type(e).__name__}: {e}"}
        return self._fabric

    # ---- what torch.distributed saw ---------------------------------------------------------------------------------------------
    def devices(self):
        torch, td = self.torch, self.td
        props = torch.cuda.get_device_properties(self.dev_index)
        mine = {"rank": self.rank, "device": self.dev_index, "name": props.name, "gcn_arch": getattr(props, "gcnArchName", ""),
                "uuid": str(getattr(props, "uuid", "")), "hbm_gib": round(props.total_memory / 2**30, 1)}
        if self.world == 1:
            return [mine]
        gathered = [None] * self.world
        td.all_gather_object(gathered, mine)
        return gathered

    # ---- one workload ----------------------------------------------------------------------------------------------------------------
    def run(self, name, steps, warmup, pre_steps, metric_override=None, shard="base", verify=False, with_cpu=False, warm_queries=0, pre_shard="auto", defer_cpu=False):
        torch, td, eng, world, rank = self.torch, self.td, self.eng, self.world, self.rank
        from neighborhoodwatch_amd.distributed import query_sharded_search, shard_bounds, sharded_search
        Q, B, d, k, metric, dtype, desc = WORKLOADS[name]
        if metric_override and metric != "maxsim":
            metric = metric_override
            desc += f" [metric overridden: {metric}]"
        peak = MFMA_PEAK_TFLOPS[dtype]
        by_query = shard == "queries" and world > 1
        lo, hi = (0, B) if by_query else shard_bounds(B, world, rank)
        maxsim = metric == "maxsim"
        if maxsim:                                                    # token matrices: docs [B, 64, d], queries [Q, 32, d]
            base_raw = eng.generate(SEED_BASE, lo * 64, (hi - lo) * 64, d, dtype).view(hi - lo, 64, d)
            queries = eng.generate(SEED_QUERY, 0, Q * 32, d, dtype).view(Q, 32, d)
        else:
            base_raw = eng.generate(SEED_BASE, lo, hi - lo, d, dtype)     # this rank's shard, raw row-major in HBM (an fp16 base selects the fp16 path)
            queries = eng.generate(SEED_QUERY, 0, Q, d, dtype)            # replicated
        host_d = torch.empty((Q, k), dtype=torch.float32).pin_memory() if rank == 0 else None
        host_i = torch.empty((Q, k), dtype=torch.int64).pin_memory() if rank == 0 else None

        ctx = {"base": base_raw, "lo": lo, "by_query": by_query}       # what the current leg searches (the pre-screened leg may re-shard)

        def search_shard(qs, kk, offset):
            if maxsim:
                eng.maxsim_set_docs(ctx["base"])
                score, doc = eng.maxsim_search(qs, kk, doc_offset=offset)
                return -score, doc                                    # ranked like distances through the shared merge
            eng.set_base(ctx["base"])                                 # layout + norms are part of every job
            return eng.search(qs, kk, metric, index_offset=offset)

        def fence():
            if world > 1:
                td.barrier()
            torch.cuda.synchronize()

        def measure(n_steps, n_warmup):
            """n_warmup untimed + n_steps timed steps, fenced on both sides; MAX over ranks of the elapsed time, SUM of the kernel accounting."""
            agg = {"tile_ms": 0.0, "tile_flops": 0.0, "tile_launches": 0, "compact_ms": 0.0, "prep_ms": 0.0, "rescan_launches": 0,
                   "overflowed_queries": 0, "rescore_ms": 0.0, "prescreen": 0, "prescreen_checked": 0, "prescreen_violations": 0,
                   "prescreen_fallbacks": 0, "walk_calibrated": 0, "walk_query_fastest": 0, "walk_ms_base_fastest": 0.0, "walk_ms_query_fastest": 0.0}
            worst = [0.0]
            coll = {}
            last = None

            def step(record, qs):
                tm = coll if record else None
                if ctx["by_query"]:
                    dist, idx = query_sharded_search(lambda q_, kk: search_shard(q_, kk, 0), qs, k, None, tm)
                else:
                    dist, idx = sharded_search(search_shard, eng.merge, qs, k, ctx["lo"], None, tm, self.args.merge)
                if rank == 0:
                    host_d[:dist.shape[0]].copy_(dist, non_blocking=True)
                    host_i[:idx.shape[0]].copy_(idx, non_blocking=True)
                torch.cuda.synchronize()
                if record:
                    s = eng.stats()
                    for key in agg:
                        agg[key] += s[key]
                    worst[0] = max(worst[0], s["prescreen_worst"])
                return dist, idx

            for _ in range(n_warmup):                                  # (a warm-up on a slice of the queries, where the caller asks for one, is said so in the line)
                step(False, queries[:warm_queries] if warm_queries else queries)
            fence()
            progress(f"{name}: {n_warmup} warm-up step(s) done, timing {n_steps}")
            t0 = time.perf_counter()
            t_said = t0
            for i_step in range(n_steps):
                last = step(True, queries)
                if time.perf_counter() - t_said > 45.0:                # (host clock only: nothing is added to the stream)
                    t_said = time.perf_counter()
                    progress(f"{name}: step {i_step + 1} of {n_steps}")
            fence()
            agg["prescreen_worst"] = worst[0]
            elapsed = time.perf_counter() - t0
            kernel = eng.tile_kernel_name()
            my_search_ms = agg["tile_ms"] + agg["compact_ms"] + agg["prep_ms"] + agg["rescore_ms"]      # this rank's kernel time over the timed steps
            if world > 1:
                t = torch.tensor([elapsed], dtype=torch.float64, device=self.red_dev)
                td.all_reduce(t, op=td.ReduceOp.MAX)
                elapsed = float(t.item())
                fl = torch.tensor([agg["tile_flops"], agg["tile_ms"]], dtype=torch.float64, device=self.red_dev)
                td.all_reduce(fl, op=td.ReduceOp.SUM)
                sum_flops, sum_ms = float(fl[0]), float(fl[1])
                cm = torch.tensor([coll.get("allgather_ms", 0.0), coll.get("merge_ms", 0.0), coll.get("slices_ms", 0.0)], dtype=torch.float64, device=self.red_dev)
                td.all_reduce(cm, op=td.ReduceOp.MAX)
                per_rank = [None] * world
                td.all_gather_object(per_rank, round(my_search_ms / n_steps, 3))
                coll = {"kind": "torch.distributed (one process per GPU)", "backend": self.backend,
                        "sharding": "query rows (base replicated): one all-gather of the [Q/N,k] slices, nothing to merge" if ctx["by_query"] else
                                    f"base rows: all-gather of the [Q,k] partial lists, merge {'of its query slice on every rank + all-gather of the slices' if self.args.merge == 'slice' else 'of all rows on every rank'}",
                        "allgather_ms": float(cm[0]) / n_steps, "merge_ms": float(cm[1]) / n_steps, "slices_allgather_ms": float(cm[2]) / n_steps,
                        "rank_kernel_ms_per_step": per_rank,             # shard searches, rank by rank: the load balance behind the max-over-ranks time
                        "allgather_bytes_per_rank": coll.get("allgather_bytes", 0),
                        "timing": "per step, max over ranks; all-gather: HIP events on the rank's current stream (RCCL) or host clock "
                                  "around the host-staged gloo exchange; merge: host clock around the blocking merge call"}
            else:
                sum_flops, sum_ms = agg["tile_flops"], agg["tile_ms"]
                coll = None
            return elapsed, agg, sum_flops, sum_ms, last, kernel, coll

        def roofline(agg_, flops_, ms_, n_steps, kernel, pk, traffic_rec):
            traffic, traffic_kernel = traffic_rec
            # a recorded figure is only quoted for the kernel it was measured on (a later build may run another one)
            if traffic is not None and kernel.split("<")[0] not in traffic_kernel:
                traffic = None
            ach = (flops_ / 1e12) / (ms_ / 1e3) if ms_ > 0 else 0.0      # FLOPs summed over ranks / kernel time summed over ranks = per-GPU rate
            # one launch per search is the dense first chunk (cap/2 base rows, a different template instance in the
            # rocprofv3 stats, ~1 ms): the per-launch figures are over the filtered launches that carry the work
            n_l = max(agg_["tile_launches"] - n_steps, 1)
            bare = self.ceiling("f16" if pk > 1000 else "f32") if world == 1 else {}
            extra = ({"bare_loop_tflops": bare["tflops"], "bare_loop_clock_ghz": bare["clock_ghz"], "frac_of_bare_loop": ach / bare["tflops"],
                      "bare_loop_note": "nwknn_mfma_ceiling on this device: the same MFMA instruction back to back on random operands in registers, "
                                        "same register tile and waves per SIMD, no operand traffic, no epilogue"}
                     if bare.get("tflops") else ({"bare_loop_error": bare["error"]} if "error" in bare else {}))
            if world == 1:
                extra.update(self.fabric())
            if agg_.get("walk_calibrated"):
                n_c = agg_["walk_calibrated"]
                extra["walk"] = {"calibrated_searches": n_c, "query_fastest_chosen": agg_["walk_query_fastest"],
                                 "ms_base_fastest_8x8": agg_["walk_ms_base_fastest"] / n_c, "ms_query_fastest_2x32": agg_["walk_ms_query_fastest"] / n_c,
                                 "note": "two full-size chunks of every search are timed, one under each super-tile walk; the rest of the walk runs the faster (nwknn_stats.walk_*)"}
            return {"bound": "mfma", "achieved": ach, "peak": pk, "unit": "TFLOP/s", "frac": ach / pk, **extra, "traffic": traffic,
                    "traffic_note": "bytes per launch, rocprofv3 PMC passes recorded in profiles/ (not sampled live; quoted only when recorded on this kernel)",
                    "kernel": kernel, "launches": n_l, "avg_launch_ms": agg_["tile_ms"] / n_l, "flops_per_launch": agg_["tile_flops"] / n_l}

        def invariant(agg_):
            """The pre-screen's checked invariant over the timed steps (nwknn_stats.prescreen_*): every rescored pair's fp16 estimate held
            against its canonical value and the bound it was pruned under."""
            return {"checked_pairs": agg_["prescreen_checked"], "violations": agg_["prescreen_violations"], "fallbacks": agg_["prescreen_fallbacks"],
                    "worst_error_over_bound": agg_["prescreen_worst"]}

        def hbm_roofline(agg_, n_steps):
            """The HBM-bound kernel of the path: the layout + norm pass that streams the raw rows (base shard and queries) once and writes
            the MFMA-ready copy.  Algorithmic bytes = rows x d x (element in + element out) + 4 per row of norms, over its HIP-event time."""
            es = 2 if dtype == "f16" else 4
            rows = (hi - lo) + Q
            byts = float(rows) * d * 2 * es * flop_rows + 4.0 * rows * flop_rows
            ms = agg_["prep_ms"] / n_steps
            ach = byts / 1e9 / (ms / 1e3) if ms > 0 else 0.0
            return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "kernel": "nwk_prep_tiled_kernel", "bytes_per_step": byts, "ms_per_step": ms,
                    "what": "streaming read of the raw base shard + queries, write of the prepared layout and norms (per step, HIP events)"}

        flop_rows = (64 if maxsim else 1)                                     # token rows per base item
        elapsed, agg, sum_flops, sum_ms, last, kernel, coll = measure(steps, warmup)         # exactly K timed steps after W warm-up steps
        flop_mult = 32 * 64 if maxsim else 1

        # the same job through the fp16 pre-screen + fp32 rescoring (engine option "prescreen"): results must be the same bits
        pre = None
        if dtype == "f32" and pre_steps > 0 and not any("prescreen" in o for o in self.args.opt):
            direct = (last[0].clone(), last[1].clone()) if rank == 0 else None
            # N > 1: the pre-screened walk has as many launches per shard as per whole base (its early, survivor-dense chunks do not shrink
            # with the shard: DESIGN.md section 7), so this leg shards the QUERY rows when the whole base and its fp16 copy fit one GPU --
            # no merge, linear in N.  --shard-prescreen base keeps the base sharding of the headline leg.
            fits = float(B) * d * 10 < 0.8 * torch.cuda.get_device_properties(self.dev_index).total_memory
            want_q = world > 1 and not ctx["by_query"] and (pre_shard == "queries" or (pre_shard == "auto" and fits))
            if want_q:
                ctx["base"] = None
                del base_raw
                torch.cuda.empty_cache()
                base_raw = ctx["base"] = eng.generate(SEED_BASE, 0, B, d, dtype)
                ctx["lo"], ctx["by_query"] = 0, True
            eng.set_prescreen(True)
            p_warm = min(warmup, 1)
            p_elapsed, p_agg, p_flops, p_ms, p_last, p_kernel, p_coll = measure(pre_steps, p_warm)
            eng.set_prescreen(False)
            if rank == 0:
                pre = {
                    "what": "same search, same bits out: fp16 MFMA pre-screen over scaled fp16 copies + canonical fp32 rescoring of every listed pair",
                    "value": Q * pre_steps / p_elapsed, "unit": "queries/s", "steps": pre_steps, "warmup": p_warm, "ms_per_step": p_elapsed / pre_steps * 1e3,
                    "pairwise_tflops": 2.0 * Q * B * d * flop_mult * pre_steps / p_elapsed / 1e12,
                    "identical_to_direct": bool(torch.equal(direct[0], p_last[0])) and bool(torch.equal(direct[1], p_last[1])),
                    "prescreen_searches": p_agg["prescreen"], "invariant": invariant(p_agg), "collective": p_coll,
                    "roofline": roofline(p_agg, p_flops, p_ms, pre_steps, p_kernel, MFMA_PEAK_TFLOPS["f16"],
                                         recorded_traffic(name + "_prescreen") if world == 1 else (None, None)),
                    "breakdown_ms_per_step": {"tile": p_agg["tile_ms"] / pre_steps, "rescore": p_agg["rescore_ms"] / pre_steps,
                                              "compact": p_agg["compact_ms"] / pre_steps, "prep": p_agg["prep_ms"] / pre_steps,
                                              "rescan_launches": p_agg["rescan_launches"], "overflowed_queries": p_agg["overflowed_queries"]},
                }

        out = None
        if rank == 0:
            out = {
                "value": Q * steps / elapsed, "unit": "queries/s", "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
                "dtype": dtype, "pairwise_tflops": 2.0 * Q * B * d * flop_mult * steps / elapsed / 1e12,
                "config": {"workload": desc, "queries": Q, "base_rows": B, "dim": d, "k": k, "metric": metric,
                           "parallelism": (f"query rows sharded x{world}, base replicated + RCCL all-gather of the [Q/N,k] slices" if by_query else
                                           f"base rows sharded x{world}" + (" + RCCL all-gather of [Q,k] partials + merge" if world > 1 else ""))},
                "roofline": roofline(agg, sum_flops, sum_ms, steps, kernel, peak, recorded_traffic(name) if world == 1 else (None, None)),
                "warmup_note": (f"the warm-up step searched the first {warm_queries} queries only (kernels, allocator and clocks warm; "
                                f"the timed steps grow the query-side buffers once)") if warm_queries else None,
                "breakdown_ms_per_step": {"tile": agg["tile_ms"] / steps, "compact": agg["compact_ms"] / steps,
                                          "prep": agg["prep_ms"] / steps, "rescore": agg["rescore_ms"] / steps, "rescan_launches": agg["rescan_launches"],
                                          "overflowed_queries": agg["overflowed_queries"]},
            }
            if world == 1 and not maxsim and not agg["prescreen"]:
                out["roofline"]["hbm"] = hbm_roofline(agg, steps)         # direct path: prep_ms is the layout pass alone
            if agg["prescreen"]:
                out["invariant"] = invariant(agg)
            if out["warmup_note"] is None:
                del out["warmup_note"]
            if coll is not None:
                out["collective"] = coll
            if pre is not None:
                out["exact_prescreen"] = pre
            if with_cpu and not maxsim:
                out["cpu_baseline"] = cpu_baseline(base_raw, queries, k, metric, B, defer=defer_cpu)
        ok = None
        if verify and rank == 0:
            if maxsim:
                eng.maxsim_set_docs(eng.generate(SEED_BASE, 0, B * 64, d, dtype).view(B, 64, d))
                rs, ri = eng.maxsim_search(queries, k)
                rd = -rs
            else:
                full = eng.generate(SEED_BASE, 0, B, d, dtype)      # same dtype as the shards: an fp16 base selects the fp16 path again
                eng.set_base(full)
                rd, ri = eng.search(queries, k, metric)
            ok = bool(torch.equal(ri, last[1])) and bool(torch.equal(rd, last[0]))
        ctx.clear()
        del base_raw, queries, last
        torch.cuda.empty_cache()
        return out, ok


def main_inlib(args):
    """--collective inlib: the same job through the C ABI's own collective -- ONE process, N GPUs (nwknn_comm_create over ncclCommInitAll,
    nwknn_search_sharded: a host thread per device for the shard searches, one ncclAllGather per array, device g merges its slice of the
    query rows).  A step = seeded base shards generated and laid out on the devices + the sharded search with HOST queries and HOST
    results (what the C ABI hands over; the torch flavour keeps the queries resident).  No torch.distributed anywhere."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if int(os.environ.get("RANK", "0")) != 0:
        return 0                                                      # started under torchrun by habit: rank 0 drives all the GPUs
    import torch
    from neighborhoodwatch_amd import KnnEngine, ShardedKnnEngine
    n = args.gpus
    have = torch.cuda.device_count()
    # rehearsal: a stand-in transport whose ranks share GPUs (tests/fake_rccl) under a DEVELOPER build of the library (NWKNN_LIB_PATH =
    # ab/libnwknn_devcomm.so, `python -m neighborhoodwatch_amd.build --dev-comm`); the shipped library refuses a GPU listed twice
    virtual = have < n and bool(os.environ.get("NWKNN_RCCL_LIBRARY")) and bool(os.environ.get("NWKNN_LIB_PATH"))
    if have < n and not virtual:
        raise SystemExit(f"--gpus {n} --collective inlib needs {n} GPUs, this node shows {have} "
                         "(NWKNN_RCCL_LIBRARY=<tests/fake_rccl build> with NWKNN_LIB_PATH=ab/libnwknn_devcomm.so rehearses the N-device path on fewer)")
    dev_list = [g % max(have, 1) for g in range(n)]
    Q, B, d, k, metric, dtype, desc = WORKLOADS[args.workload]
    if metric == "maxsim":
        raise SystemExit("--collective inlib covers the kNN workloads (nwknn_search_sharded); MaxSim shards through torch.distributed")
    if args.metric:
        metric = args.metric
    with KnnEngine(0) as gen:
        queries = gen.generate(SEED_QUERY, 0, Q, d, dtype).cpu().numpy()
    eng = ShardedKnnEngine(dev_list, profile=True)
    for kvs in args.opt:
        for kv in kvs.split(","):
            name, value = kv.split("=")
            eng.set_option(name, int(value))
    agg = {"search_ms": 0.0, "allgather_ms": 0.0, "merge_ms": 0.0, "tile_flops": 0.0, "tile_ms": 0.0}

    def step(record):
        eng.generate_base(SEED_BASE, B, d, dtype)
        out = eng.search(queries, k, metric)
        if record:
            st = eng.stats()
            for key in agg:
                agg[key] += st[key]
        return out
    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dist, idx = step(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ach = agg["tile_flops"] / 1e12 / (agg["tile_ms"] / 1e3) if agg["tile_ms"] > 0 else 0.0
    peak = MFMA_PEAK_TFLOPS[dtype]
    devices = []
    for g in range(n):
        pr = torch.cuda.get_device_properties(dev_list[g])
        devices.append({"rank": g, "device": dev_list[g], "name": pr.name, "gcn_arch": getattr(pr, "gcnArchName", ""), "uuid": str(getattr(pr, "uuid", "")),
                        "hbm_gib": round(pr.total_memory / 2**30, 1)})
    line = {"metric": "queries/sec + pairwise-dist TFLOP/s, 100k x 10M d=1024 k=100" if args.workload == "c3" else f"queries/sec ({args.workload})",
            "value": Q * args.steps / elapsed, "unit": "queries/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic (seeded counter-based embeddings in [-1,1); base shards generated on the devices, queries handed over from host memory)",
            "ranks": 1, "devices": devices, "pairwise_tflops": 2.0 * Q * B * d * args.steps / elapsed / 1e12,
            "config": {"workload": desc, "queries": Q, "base_rows": B, "dim": d, "k": k, "metric": metric,
                       "parallelism": f"base rows sharded x{n} inside ONE process (C ABI: nwknn_comm_*), RCCL all-gather of [Q,k] partials + sliced merge"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                         "kernel": "per-device rate: tile FLOPs summed over the devices / tile-kernel time summed over the devices"},
            "collective": {"kind": "in-library (C ABI nwknn_search_sharded: one process, N GPUs, ncclCommInitAll)",
                           "backend": "REHEARSAL: stand-in transport, virtual devices sharing GPUs (NWKNN_RCCL_LIBRARY + developer build of the library)" if virtual else "rccl",
                           "search_ms": agg["search_ms"] / args.steps, "allgather_ms": agg["allgather_ms"] / args.steps,
                           "merge_ms": agg["merge_ms"] / args.steps, "allgather_bytes_per_rank": n * Q * k * 12,
                           "timing": "per step, HIP events, maxima over the devices (nwknn_comm_stats)"}}
    ok = None
    if args.verify:
        with KnnEngine(0) as one:
            one.set_base(one.generate(SEED_BASE, 0, B, d, dtype))
            rd, ri = one.search(queries, k, metric)
        import numpy as np
        ok = bool(np.array_equal(ri, idx)) and bool(np.array_equal(rd.view(np.uint32), dist.view(np.uint32)))
    eng.close()
    print(json.dumps(line), flush=True)
    if args.verify:
        print(json.dumps({"verify": "in-library N-device result == single-engine result", "ok": ok}), flush=True)
    return 3 if (args.verify and not ok) else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="base", choices=["base", "queries"],
                    help="N > 1: 'base' (default, the headline) shards the base rows and merges [Q,k] partials after one all-gather; "
                         "'queries' replicates the base and shards the query rows (no merge; only when the base fits one GPU)")
    ap.add_argument("--shard-prescreen", default="auto", choices=["auto", "base", "queries"],
                    help="N > 1, sharding of the pre-screened leg: 'auto' = query rows when the whole base + its fp16 copy fit one GPU, else as --shard")
    ap.add_argument("--merge", default="slice", choices=["slice", "replicated"],
                    help="N > 1, base sharding: 'slice' = rank g merges its slice of the query rows and the slices are all-gathered; 'replicated' = every rank merges all rows")
    ap.add_argument("--collective", default="torch", choices=["torch", "inlib"],
                    help="'torch' = one process per GPU over torch.distributed (RCCL); 'inlib' = ONE process driving N GPUs through the C ABI's "
                         "nwknn_comm_* / nwknn_search_sharded (RCCL inside libnwknn.so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prescreen", action="store_true", help="skip the second measurement (fp16 pre-screen + fp32 rescoring)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary block (C2, one C3 shard, the C4 per-GPU slice) that follows the default workload at N=1")
    ap.add_argument("--files-base-rows", type=int, default=REF_FILES_BASE_ROWS, help="base rows of the refcli file leg (default: the reference's 4M; tests shrink it)")
    ap.add_argument("--files", action="store_true", help="--workload refcli: also run the job through compute_knn from a synthetic parquet file (write excluded, load included)")
    ap.add_argument("--verify", action="store_true", help="compare the merged result with a single-engine search of the whole base (small workloads)")
    ap.add_argument("--metric", default=None, choices=["sqeuclidean", "one_minus_dot", "cosine"], help="override the workload's metric (dev)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="engine option (nwknn_set_option), dev use")
    args = ap.parse_args()
    if os.environ.get("NWKNN_BENCH_WATCHDOG"):                   # dev: every N seconds, every thread's Python stack on stderr (a hang names itself)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NWKNN_BENCH_WATCHDOG"]), repeat=True, file=sys.stderr)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.collective == "inlib":
        raise SystemExit(main_inlib(args))                    # one process whatever N: nothing to launch
    if args.gpus > 1 and "RANK" not in os.environ:
        why = profiler_preload()
        if why:
            print(f"bench.py: refusing to start {args.gpus} ranks from under a profiler ({why}): its preloaded library has already initialised "
                  "the GPU in this process, and starting children from here is the exec-after-GPU-init this pool forbids.  Profile ONE rank "
                  "(`rocprofv3 ... -- python3 bench.py --gpus 1`), or the per-GPU pieces (`--workload c3shard`, `--workload c4gpu`).",
                  file=sys.stderr, flush=True)
            raise SystemExit(4)
        raise SystemExit(self_launch(args.gpus))              # nothing above has imported torch or touched the GPU

    b = Bench(args)
    td, world, rank = b.td, b.world, b.rank
    devices = b.devices()
    out, ok = b.run(args.workload, args.steps, args.warmup, 0 if args.no_prescreen else min(args.steps, 5), args.metric, args.shard,
                    args.verify, with_cpu=(world == 1 and not args.no_cpu_baseline), pre_shard=args.shard_prescreen,
                    defer_cpu=(world == 1 and args.workload == "c3" and not args.no_secondary and not args.opt and not args.metric))
    if rank == 0:
        line = {"metric": "queries/sec + pairwise-dist TFLOP/s, 100k x 10M d=1024 k=100" if args.workload == "c3" else f"queries/sec ({args.workload})",
                "value": out.pop("value"), "unit": out.pop("unit"), "n_gpus": world, "steps": out.pop("steps"), "warmup": out.pop("warmup"),
                "ms_per_step": out.pop("ms_per_step"), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": out.pop("dtype"), "data": "synthetic (seeded counter-based embeddings in [-1,1), generated on device)",
                "ranks": td.get_world_size() if world > 1 else 1, "devices": devices}
        line.update(out)
    if world == 1 and args.workload == "refcli" and rank == 0:
        line["vs_reference_published"] = ref_published_block(line["value"], line.get("exact_prescreen"))
        if args.files:
            line["through_files"] = ref_cli_file_legs(b.eng, args.files_base_rows)
    if world == 1 and args.workload == "c3" and not args.no_secondary and not args.opt and not args.metric:
        sec = {}
        # the file legs' source files (29 GB of parquet) are written by a background thread from here on, under the c4gpu leg
        # (it costs the main thread nothing; whether its ~20 s of compute legs still fit is decided when the refcli leg gets there)
        ref_files = RefFiles(b.eng, args.files_base_rows)
        for name, s_steps, s_pre, s_warm_q, est_s in SECONDARY:
            t0 = time.perf_counter()
            if t0 - T_PROCESS_START + est_s > SECONDARY_BUDGET_S:
                sec[name] = {"skipped": f"time box: {t0 - T_PROCESS_START:.0f} s into the run, this leg needs ~{est_s} s and the block stops at "
                                        f"{SECONDARY_BUDGET_S:.0f} s; run `python bench.py --workload {name}` for it"}
                continue
            try:
                if name != "c4gpu":
                    # Only the c4gpu leg (a handful of 0.1-0.8 s launches per search: the host idles) shares the machine with the background
                    # threads.  The other legs submit a launch every few milliseconds, and this container's CPU quota (16 CPUs' worth; the
                    # BLAS baseline alone takes all of it) freezes EVERY thread of the process, the submitting one included, once it is spent:
                    # a refcli leg that overlapped the baseline read 5 033 q/s instead of 11 600 (profiles/r05bf_*).  So: wait for them here.
                    if hasattr(line.get("cpu_baseline"), "result"):
                        try:
                            line["cpu_baseline"] = line["cpu_baseline"].result()
                            line["cpu_baseline"]["ran"] = "on a background thread under the c4gpu leg (GPU work with an idle host); the later legs waited for it"
                        except Exception as e:
                            line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
                    if ref_files is not None:
                        ref_files.thread.join()
                    t0 = time.perf_counter()
                progress(f"secondary leg {name}")
                s_out, _ = b.run(name, s_steps, 1, 0 if args.no_prescreen else s_pre, warm_queries=s_warm_q)
                if name == "refcli":
                    s_out["vs_reference_published"] = ref_published_block(s_out["value"], s_out.get("exact_prescreen"))
                    if ref_files is not None and time.perf_counter() - T_PROCESS_START + REF_FILES_EST_S <= SECONDARY_BUDGET_S:
                        progress("secondary leg refcli: through files")
                        s_out["through_files"] = ref_files.run(b.eng)
                        ref_files = None
                    else:
                        s_out["through_files"] = {"skipped": f"time box ({REF_FILES_EST_S} s needed); run `python bench.py --workload refcli --files`"}
                s_out["wall_s"] = time.perf_counter() - t0
                sec[name] = s_out
            except Exception as e:                                  # a secondary figure must never cost the headline line
                sec[name] = {"error": f"{type(e).__name__}: {e}"}
        if ref_files is not None:
            ref_files.discard()
        line["secondary"] = sec
        if hasattr(line.get("cpu_baseline"), "result"):               # the host-side baseline ran on a background thread under the GPU-only legs above
            try:
                line["cpu_baseline"] = line["cpu_baseline"].result()
                line["cpu_baseline"]["ran"] = "on a background thread while the secondary legs ran"
            except Exception as e:
                line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        line["wall_s_since_process_start"] = time.perf_counter() - T_PROCESS_START
    if rank == 0:
        print(json.dumps(line), flush=True)
        if args.verify:
            print(json.dumps({"verify": "merged N-rank result == single-engine result", "ok": ok}), flush=True)
    rc = 3 if (args.verify and rank == 0 and not ok) else 0
    if world > 1:
        td.barrier()
        td.destroy_process_group()
    if rc:
        raise SystemExit(rc)


if __name__ == "__main__":
    main()
