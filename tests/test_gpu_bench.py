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
    cb = b["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "queries/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    pre = b["exact_prescreen"]
    assert pre["identical_to_direct"] is True and pre["value"] > 0 and pre["roofline"]["peak"] == 2500.0


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
