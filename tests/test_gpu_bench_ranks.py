"""bench.py's N > 1 path end to end on ONE GPU: two ranks (gloo for the all-reduce, both on cuda:0)
render their iteration stripes, combine, and the image equals a single rank rendering every
iteration; the JSON contract of the bench line is checked on the way."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu():
    env = dict(os.environ, VPT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--width", "320", "--height", "180", "--spp", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["unit"] == "Msamples/s"
    assert d["value"] > 0 and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["spp_per_gpu"] == 4


def test_striped_ranks_equal_single_rank(pkg):
    """the image of 2 striped ranks + combine_means == 1 rank with twice the iterations"""
    import torch
    sd = pkg.scene.dragon_scene(160, 90, "sun")
    one = pkg.scene.HipBinding(sd, device=0)
    one.render(8)
    one.sync()
    parts = []
    for rank in range(2):
        first, stride, pre = pkg.dist.stripe(rank, 2)
        hb = pkg.scene.HipBinding(sd, device=0)
        if pre:
            hb.ctx.blue_noise_advance(hb.blue_noise, pre, sd.width * sd.height)
        hb.render(4, iter_stride=stride, iteration=first)
        hb.sync()
        parts.append(hb.accum.clone())
    combined = (parts[0] * 4 + parts[1] * 4) / 8
    np.testing.assert_allclose(combined.cpu().numpy(), one.accum.cpu().numpy(), rtol=2e-5, atol=1e-7)


def test_single_rank_bench_line_contract():
    """the N = 1 bench line: every key of the contract, the roofline object and the CPU baseline leg"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--width", "320", "--height", "180",
           "--spp", "4", "--cpu-iters", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
