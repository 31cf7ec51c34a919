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
