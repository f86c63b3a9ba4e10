"""bench.py prints exactly one JSON line with the contract's fields (pytest -m gpu; small workload so it runs in
seconds: EDSR-baseline, 2 timed steps, batch 4)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--batch", "4", "--workload", "edsr"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["unit"] == "patches/s" and d["value"] > 0 and abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py spawns its two ranks as a child
    torch.distributed.run (never exec) and relays rank 0's single JSON line.  Rehearsed on the one GPU of this box: both ranks
    on cuda:0, gloo transport; the default N > 1 workload = BASELINE config 4 (QRCAN, global batch 32, strong scaling,
    hipGraph replay), with the weak-scaling object and the watchdogged overlapped-exchange object beside it."""
    env = dict(os.environ, SISR_BENCH_SHARE_GPU="1", SISR_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 32 and d["config"]["per_gpu_batch"] == 16 and d["value"] > 0
    assert "grad_exchange" in d["config"] and d["config"]["grad_exchange"]
    w = d["weak_scaling"]
    assert w.get("per_gpu_batch") == 32 and w.get("global_batch") == 64 and w["value"] > 0, w
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_bench_refuses_a_world_size_mismatch():
    """Inside a launcher's world the --gpus flag must agree with it."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)
