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
  secondary    -- N=1, default workload only: the other BASELINE configs one GPU can hold (C2; one of eight C3 shards; the C4 per-GPU
                  slice), a few steps each with their own value / ms_per_step / roofline, on the same clock as the headline.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WORKLOADS = {
    # name: (Q, B, d, k, metric, dtype, description)
    "c3": (100_000, 10_000_000, 1024, 100, "sqeuclidean", "f32", "C3: 100k queries x 10M base, d=1024 (e5-large-v2 shape), k=100, fp32 sqeuclidean"),
    "c2": (10_000, 1_000_000, 768, 100, "sqeuclidean", "f32", "C2: 10k queries x 1M base, d=768, k=100, fp32 sqeuclidean"),
    "dev": (4_096, 262_144, 256, 100, "sqeuclidean", "f32", "dev: 4k x 256k, d=256, k=100 (not a headline number)"),
    # what ONE of eight GPUs does per step of the headline job at --gpus 8 (all queries x its 1.25M-row shard; no collective):
    # 8 x its queries/s is the ceiling of the 8-GPU line, measurable on a one-GPU box
    "c3shard": (100_000, 1_250_000, 1024, 100, "sqeuclidean", "f32", "one of eight C3 shards: 100k queries x 1.25M base rows, d=1024, k=100, fp32 (not a headline number)"),
    # BASELINE config 4 (fp16 MFMA + L2, 8 GPUs) and the piece of it one GPU sees (its 12.5M-row shard, a tenth of the queries)
    "c4": (1_000_000, 100_000_000, 1536, 100, "sqeuclidean", "f16", "C4: 1M queries x 100M base, d=1536 (ada-002 shape), k=100, fp16 MFMA sqeuclidean"),
    "c4gpu": (100_000, 12_500_000, 1536, 100, "sqeuclidean", "f16", "C4 per-GPU slice: 100k queries x 12.5M base rows (one of 8 shards), d=1536, k=100, fp16"),
    # BASELINE config 5: MaxSim late interaction (an extension -- the reference's `ck` is flat token kNN): Q = queries of 32 tokens,
    # B = docs of 64 tokens, d = 128, k = 1000; FLOPs = 2 * (32 Q) * (64 B) * d
    "c5": (2_048, 1_000_000, 128, 1000, "maxsim", "f32", "C5: MaxSim, 2048 queries x 32 tokens vs 1M docs x 64 tokens, d=128, k=1000, fp32"),
    "dev5": (256, 100_000, 128, 100, "maxsim", "f32", "dev5: MaxSim 256 queries vs 100k docs (not a headline number)"),
    "dev16": (32_768, 1_048_576, 1536, 100, "sqeuclidean", "f16", "dev16: 32k x 1M, d=1536, k=100, fp16 (not a headline number)"),
}
SECONDARY = (("c2", 10, 2), ("c3shard", 3, 2), ("c4gpu", 2, 0))     # (workload, timed steps, timed steps of its pre-screened leg); 1 warm-up each
SEED_BASE, SEED_QUERY = 0x6E770001, 0x6E770002
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0}      # MI355X_MICROARCH.md: dense f32 / f16 MFMA peaks


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


def cpu_baseline(base_dev, queries_dev, k, metric, full_b, flop_budget=0.7e13):
    """Brute force on the host cores over a bounded sample of the same workload: numpy/BLAS (the oracle's sgemm form) and, beside it,
    scikit-learn's NearestNeighbors(algorithm='brute').  Each leg is sized to ~flop_budget of sgemm (10-20 s on a 16..64-core host)."""
    import numpy as np
    from oracle import cpu_oracle
    cores = len(os.sched_getaffinity(0))
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [cores])
    except Exception:
        blas_threads = cores
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
        "value": qs / dt * scale, "unit": "queries/s", "cores": int(blas_threads), "kind": "port",
        "sample": f"{qs} queries x first {bs} of {full_b} base rows in {dt:.2f}s (numpy sgemm + argpartition, "
                  f"oracle.cpu_oracle.knn_blas_fp32); rate scaled by {bs}/{full_b} to the full base",
    }
    if metric in ("sqeuclidean", "cosine"):
        try:
            from sklearn.neighbors import NearestNeighbors
            nn = NearestNeighbors(n_neighbors=min(k, bs), algorithm="brute", metric=metric, n_jobs=-1)
            nn.fit(base[: min(bs, 20000)]).kneighbors(qry[:64], return_distance=True)     # warm-up
            t0 = time.perf_counter()
            nn.fit(base).kneighbors(qry, return_distance=True)
            dts = time.perf_counter() - t0
            out["sklearn"] = {"value": qs / dts * scale, "unit": "queries/s", "cores": cores, "kind": "library",
                              "sample": f"sklearn.neighbors.NearestNeighbors(algorithm='brute', metric='{metric}', n_jobs=-1).kneighbors on the same "
                                        f"{qs} x {bs} sample in {dts:.2f}s; same scaling"}
        except Exception as e:                                                    # the baseline is a report, never a reason to lose the line
            out["sklearn"] = {"error": f"{type(e).__name__}: {e}"}
    return out


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
    def run(self, name, steps, warmup, pre_steps, metric_override=None, shard="base", verify=False, with_cpu=False):
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

        def search_shard(qs, kk, offset):
            if maxsim:
                eng.maxsim_set_docs(base_raw)
                score, doc = eng.maxsim_search(qs, kk, doc_offset=offset)
                return -score, doc                                    # ranked like distances through the shared merge
            eng.set_base(base_raw)                                    # layout + norms are part of every job
            return eng.search(qs, kk, metric, index_offset=offset)

        def fence():
            if world > 1:
                td.barrier()
            torch.cuda.synchronize()

        def measure(n_steps, n_warmup):
            """n_warmup untimed + n_steps timed steps, fenced on both sides; MAX over ranks of the elapsed time, SUM of the kernel accounting."""
            agg = {"tile_ms": 0.0, "tile_flops": 0.0, "tile_launches": 0, "compact_ms": 0.0, "prep_ms": 0.0, "rescan_launches": 0,
                   "overflowed_queries": 0, "rescore_ms": 0.0, "prescreen": 0}
            coll = {}
            last = None

            def step(record):
                tm = coll if record else None
                if by_query:
                    dist, idx = query_sharded_search(lambda qs, kk: search_shard(qs, kk, 0), queries, k, None, tm)
                else:
                    dist, idx = sharded_search(search_shard, eng.merge, queries, k, lo, None, tm)
                if rank == 0:
                    host_d.copy_(dist, non_blocking=True)
                    host_i.copy_(idx, non_blocking=True)
                torch.cuda.synchronize()
                if record:
                    s = eng.stats()
                    for key in agg:
                        agg[key] += s[key]
                return dist, idx

            for _ in range(n_warmup):
                step(False)
            fence()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                last = step(True)
            fence()
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
                cm = torch.tensor([coll.get("allgather_ms", 0.0), coll.get("merge_ms", 0.0)], dtype=torch.float64, device=self.red_dev)
                td.all_reduce(cm, op=td.ReduceOp.MAX)
                per_rank = [None] * world
                td.all_gather_object(per_rank, round(my_search_ms / n_steps, 3))
                coll = {"backend": self.backend, "allgather_ms": float(cm[0]) / n_steps, "merge_ms": float(cm[1]) / n_steps,
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
            return {"bound": "mfma", "achieved": ach, "peak": pk, "unit": "TFLOP/s", "frac": ach / pk, "traffic": traffic,
                    "traffic_note": "bytes per launch, rocprofv3 PMC passes recorded in profiles/ (not sampled live; quoted only when recorded on this kernel)",
                    "kernel": kernel, "launches": n_l, "avg_launch_ms": agg_["tile_ms"] / n_l, "flops_per_launch": agg_["tile_flops"] / n_l}

        elapsed, agg, sum_flops, sum_ms, last, kernel, coll = measure(steps, warmup)         # exactly K timed steps after W warm-up steps
        flop_mult = 32 * 64 if maxsim else 1

        # the same job through the fp16 pre-screen + fp32 rescoring (engine option "prescreen"): results must be the same bits
        pre = None
        if dtype == "f32" and pre_steps > 0 and not any("prescreen" in o for o in self.args.opt):
            direct = (last[0].clone(), last[1].clone()) if rank == 0 else None
            eng.set_prescreen(True)
            p_warm = min(warmup, 1)
            p_elapsed, p_agg, p_flops, p_ms, p_last, p_kernel, _ = measure(pre_steps, p_warm)
            eng.set_prescreen(False)
            if rank == 0:
                pre = {
                    "what": "same search, same bits out: fp16 MFMA pre-screen over scaled fp16 copies + canonical fp32 rescoring of every listed pair",
                    "value": Q * pre_steps / p_elapsed, "unit": "queries/s", "steps": pre_steps, "warmup": p_warm, "ms_per_step": p_elapsed / pre_steps * 1e3,
                    "pairwise_tflops": 2.0 * Q * B * d * flop_mult * pre_steps / p_elapsed / 1e12,
                    "identical_to_direct": bool(torch.equal(direct[0], p_last[0])) and bool(torch.equal(direct[1], p_last[1])),
                    "prescreen_searches": p_agg["prescreen"],
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
                "breakdown_ms_per_step": {"tile": agg["tile_ms"] / steps, "compact": agg["compact_ms"] / steps,
                                          "prep": agg["prep_ms"] / steps, "rescore": agg["rescore_ms"] / steps, "rescan_launches": agg["rescan_launches"],
                                          "overflowed_queries": agg["overflowed_queries"]},
            }
            if coll is not None:
                out["collective"] = coll
            if pre is not None:
                out["exact_prescreen"] = pre
            if with_cpu and not maxsim:
                out["cpu_baseline"] = cpu_baseline(base_raw, queries, k, metric, B)
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
        del base_raw, queries, last
        torch.cuda.empty_cache()
        return out, ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="base", choices=["base", "queries"],
                    help="N > 1: 'base' (default, the headline) shards the base rows and merges [Q,k] partials after one all-gather; "
                         "'queries' replicates the base and shards the query rows (no merge; only when the base fits one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prescreen", action="store_true", help="skip the second measurement (fp16 pre-screen + fp32 rescoring)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary block (C2, one C3 shard, the C4 per-GPU slice) that follows the default workload at N=1")
    ap.add_argument("--verify", action="store_true", help="compare the merged result with a single-engine search of the whole base (small workloads)")
    ap.add_argument("--metric", default=None, choices=["sqeuclidean", "one_minus_dot", "cosine"], help="override the workload's metric (dev)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="engine option (nwknn_set_option), dev use")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))              # nothing above has imported torch or touched the GPU

    b = Bench(args)
    td, world, rank = b.td, b.world, b.rank
    devices = b.devices()
    out, ok = b.run(args.workload, args.steps, args.warmup, 0 if args.no_prescreen else min(args.steps, 5), args.metric, args.shard,
                    args.verify, with_cpu=(world == 1 and not args.no_cpu_baseline))
    if rank == 0:
        line = {"metric": "queries/sec + pairwise-dist TFLOP/s, 100k x 10M d=1024 k=100" if args.workload == "c3" else f"queries/sec ({args.workload})",
                "value": out.pop("value"), "unit": out.pop("unit"), "n_gpus": world, "steps": out.pop("steps"), "warmup": out.pop("warmup"),
                "ms_per_step": out.pop("ms_per_step"), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": out.pop("dtype"), "data": "synthetic (seeded counter-based embeddings in [-1,1), generated on device)",
                "ranks": td.get_world_size() if world > 1 else 1, "devices": devices}
        line.update(out)
    if world == 1 and args.workload == "c3" and not args.no_secondary and not args.opt and not args.metric:
        sec = {}
        for name, s_steps, s_pre in SECONDARY:
            t0 = time.perf_counter()
            try:
                s_out, _ = b.run(name, s_steps, 1, 0 if args.no_prescreen else s_pre)
                s_out["wall_s"] = time.perf_counter() - t0
                sec[name] = s_out
            except Exception as e:                                  # a secondary figure must never cost the headline line
                sec[name] = {"error": f"{type(e).__name__}: {e}"}
        line["secondary"] = sec
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
