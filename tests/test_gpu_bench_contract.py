"""The one JSON line bench.py prints: every key the harness reads, with the types and ranges it
expects, from a short run on a two-layer cut of the benchmark shape (the full shape only differs in
how long it takes)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_every_field_of_the_contract(mv):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--layers", "2", "--steps", "3",
           "--warmup", "1", "--tail-steps", "2"]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    d = json.loads(lines[0])

    assert d["metric"].startswith("decode tokens/sec") and d["unit"] == "tokens/s"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 0.02, "value is tokens of the whole job per second"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong")
    assert d["vs_baseline"] is None and d["dtype"] == "bf16" and "synthetic" in d["data"]
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]

    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["avg_launch_us"] > 0 and r["launches_per_step"] == 4 * 2 + 1
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["avg_launch_us"] / 1e3) / r["achieved"] < 0.01

    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["unit"] == "tokens/s" and c["cores"] >= 1 and c["kind"] in ("port", "reference")
    assert isinstance(c["sample"], str) and c["sample"]

    for leg in (d["bs32"]["even"], d["bs32"]["ragged"]):
        assert leg["value"] > 0 and leg["batch"] == 32 and leg["roofline"]["kernel"] == "wgemm_mfma_kernel"
    assert d["prefill"]["value"] > 0 and d["prefill"]["bound"] == "mfma"
    assert d["engine_bs32"]["value"] > 0 and d["engine_bs32"]["generated"] == 32 * 128


def test_bench_starts_its_own_ranks_when_asked_for_several_gpus(mv):
    """`bench.py --gpus 2` without a launcher: the process starts the two ranks itself (here both on
    the one card, which MVLLM_SHARE_GPU=1 permits) and passes rank 0's single line through; the line
    says two GPUs, which reducer ran and how the LM head was laid out."""
    env = dict(os.environ, MVLLM_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "5",
           "--warmup", "2", "--tail-steps", "2"]
    run = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280, cwd=ROOT,
                         env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "strong"
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 0.02
    assert d["config"]["parallelism"] == "tp2" and isinstance(d["config"]["lm_head"], str)
    a = d["allreduce"]
    assert a["reducer"] in ("IpcAllReduce", "DistAllReduce") and a["us_per_op"] > 0
    assert a["ops_per_step"] == 4 and 0 < a["share_of_step"] < 1
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["launches_per_step"] == 4 * 2 + 1


def test_bench_refuses_more_ranks_than_gpus_unless_told_to_share(mv):
    import torch

    n = torch.cuda.device_count() + 1
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MVLLM_SHARE_GPU"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--layers", "1",
                          "--steps", "1", "--warmup", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=120, cwd=ROOT, env=env)
    assert run.returncode == 2 and not run.stdout.strip()
    assert "MVLLM_SHARE_GPU" in run.stderr
