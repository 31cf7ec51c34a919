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
    # both scalings travel in the one line (round 5): the job's rate with every rank rendering the 4 iterations, and with the 4 split over the 2 ranks
    assert d["weak"]["value"] == d["value"] and d["weak"]["spp_per_gpu"] == 4 and d["weak"]["spp_job"] == 8
    assert d["strong"]["value"] > 0 and d["strong"]["spp_per_gpu"] == 2 and d["strong"]["spp_job"] == 4 and d["strong"]["n_gpus"] == 2
    # a strong-scaling step of a few iterations per rank is one independent frame: three of them in flight per rank (three contexts dealt round robin)
    assert d["strong"]["frames_in_flight"] == 3 and d["weak"]["frames_in_flight"] == 1
    # fixed total work: the job's 4 iterations split over the 2 ranks
    r = subprocess.run(cmd + ["--scaling", "strong"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["spp_per_gpu"] == 2 and d["value"] > 0
    assert d["strong"]["value"] == d["value"] and d["weak"]["spp_per_gpu"] == 4 and d["weak"]["value"] > 0


def test_eight_ranks_strong_scaling_dry_run():
    """`bench.py --gpus 8 --scaling strong` end to end with the job's 12 iterations NOT a multiple of the ranks (2 2 2 2 1 1 1 1), eight ranks
    sharing this box's one GPU through the gloo fallback: the striping, the weighted reduce and the bench line of the 8-rank job the driver
    runs on an 8-GPU node (there with one RCCL all-reduce under the C ABI)."""
    env = dict(os.environ, VPT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
           "--width", "256", "--height", "144", "--spp", "12", "--no-cpu-baseline", "--scaling", "strong"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["spp_per_gpu"] == 2 and d["config"]["spp_job"] == 12          # rank 0 renders iterations 0 and 8
    assert abs(d["value"] - 256 * 144 * 12 / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["value"]


def test_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what a scaling run issues): bench.py re-executes itself under
    torch.distributed.run, rank 0 prints ONE JSON line with n_gpus = 2"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["VPT_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--width", "320", "--height", "180", "--spp", "4", "--scaling", "strong"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["spp_per_gpu"] == 2 and d["value"] > 0
    assert d["config"]["collective"]["comm_ranks"] == 2


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
           "--spp", "4", "--cpu-iters", "2", "--frames", "4", "--grid-scale", "0.125"]
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
    # `frac` is BASELINE.md 3's figure (reference-defined look-up counts, whole step); the tracer's own figures travel next to it
    for k in ("frac_kernel_issued_fetches", "frac_step_issued_fetches", "hbm_measured_frac", "tracer_grays_per_s", "cache_build_ms_per_view", "definition"):
        assert k in rf, k
    bs = rf["bytes_per_sample"]
    assert abs(bs["kernel_must_move"] - bs["lookup_bytes"] - bs["record_stream_bytes"]) < 0.02
    assert abs(rf["achieved"] - bs["survey_8d_reference_counts"] * d["value"] * 1e6 / 1e9) <= 2e-3 * rf["achieved"]
    assert 0 < rf["frac"] < 1 and 0 < rf["frac_kernel_issued_fetches"] < 1 and 0 < rf["frac_step_issued_fetches"] < 1
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
    # full-frame parity of the same iterations: HIP vs the CPU render (the reference's kernel where oracle/_ref exists)
    assert cb["parity_rel_l2"] <= 1e-3 and cb["parity_depth_pixels_differing"] == 0
    # the literal drop-in call and the other BASELINE configs travel in the same line
    pf = d["per_frame"]
    assert pf["value"] > 0 and pf["frames"] == 4 and pf["ms_per_frame"] > 0
    oc = d["other_configs"]
    assert len(oc) == 3
    for o in oc:
        rfo = o["roofline"]
        assert o["value"] > 0 and "workload" in o["config"] and 0 < rfo["frac_step_issued_fetches"] < 1
        # no figure in the line is above 1 under the name of a fraction (round 5): where BASELINE.md's bytes per sample x samples/s exceed the peak -- config 5: the
        # reference-defined counts charge the colour look-ups the reference evaluates and discards -- `frac` is null, `frac_void` says so, the raw ratio travels under
        # another name and the kernel's own figure is promoted
        if rfo["frac_void"]:
            assert rfo["frac"] is None and rfo["achieved"] is None and rfo["reference_count_bytes_over_peak"] > 1.0
            assert rfo["frac_promoted"]["name"] == "frac_kernel_issued_fetches" and 0 < rfo["frac_promoted"]["value"] < 1
        else:
            assert 0 < rfo["frac"] < 1
    # both scalings, the cold view, the per-frame call either way and config 1 on one host thread travel in the same line
    assert d["weak"]["value"] == d["value"] == d["strong"]["value"] and d["weak"]["frames_in_flight"] == 1
    assert 0 < rf["cold_view_msamples_per_s"] <= d["value"] * 1.001 and rf["frac_void"] is False
    assert pf["frame_by_frame"]["value"] > 0 and "frame_ahead" in pf
    c1 = d["c1_cpu_single_thread"]
    assert c1["cpu"]["cores"] == 1 and c1["cpu"]["value"] > 0 and c1["hip"]["value"] > 0 and c1["parity_rel_l2"] <= 1e-3 and c1["parity_depth_pixels_differing"] == 0


def test_frames_in_flight_through_rccl_single_rank():
    """the strong-scaling step with THREE frames in flight (three contexts, three RCCL communicators, the all-reduces chained by events) -- the N-rank code path with one
    rank (VPT_BENCH_FORCE_DIST=1: process group + communicators under the C ABI + stream-ordered reduces), which is what a 1-GPU box can run of it"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(VPT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "7", "--warmup", "2", "--width", "320", "--height", "180", "--spp", "4",
           "--scaling", "strong", "--frames-in-flight", "3", "--no-cpu-baseline", "--no-other-configs", "--no-per-frame"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["frames_in_flight"] == 3
    assert d["config"]["collective"]["backend"].startswith("rccl") and d["config"]["collective"]["comm_ranks"] == 1


def test_c_abi_allreduce_single_rank(pkg):
    """vpt_comm_* / vpt_allreduce_accum (RCCL loaded at run time, below the C ABI) on a one-rank communicator: the scale ->
    grouped all-reduce -> divide chain on the context's stream returns the rank's own mean, and vpt_resolve_display
    reproduces the display image the batch wrote.  (Two ranks cannot share the single GPU of this box: RCCL rejects
    duplicate devices; the 2-rank image identity is covered by test_striped_ranks_equal_single_rank and, through gloo,
    tests/test_dist_gloo.py; tools/vpt_cli --ranks and bench.py --gpus N run the same entry points on N GPUs.)"""
    import torch
    sd = pkg.scene.dragon_scene(160, 90, "sun")
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.ctx.comm_init(1, 0, hb.ctx.comm_unique_id())
    hb.render(5)
    hb.sync()
    before, disp = hb.accum.clone(), hb.display.clone()
    pkg.dist.combine_means(hb.accum, 5, ctx=hb.ctx)                 # C-ABI path (the context has a communicator)
    hb.display.zero_()
    torch.cuda.synchronize()
    hb.ctx.resolve_display(hb.kp)
    hb.sync()
    np.testing.assert_allclose(hb.accum.cpu().numpy(), before.cpu().numpy(), rtol=3e-7, atol=0)      # (5 x) / 5: at most one rounding each way
    d0, d1 = disp.cpu().numpy().view(np.uint8).astype(np.int32), hb.display.cpu().numpy().view(np.uint8).astype(np.int32)
    assert np.abs(d0 - d1).max() <= 1
    hb.ctx.comm_destroy()
    with pytest.raises(pkg.VptError):
        hb.ctx.allreduce_accum(hb.accum, 5)                         # no communicator any more: an error, not a silent no-op


def test_cli_two_ranks_equal_one(pkg, tmp_path):
    """the C++ host on 2 GPUs (threads, one context each, RCCL all-reduce under the C ABI) == the same job on 1 GPU"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import test_gpu_cli as tc
    d = str(tmp_path)
    tc._write_assets(pkg, d)
    imgs = []
    for ranks in (1, 2):
        out = os.path.join(d, "r%d" % ranks)
        info = tc._run(pkg, [os.path.join(d, "dragon.vdb"), "--assets", d, "--size", "160", "90", "--spp", "8", "--out", out, "--ranks", str(ranks)])
        assert info["ranks"] == ranks
        imgs.append(tc._read_pfm(out + ".pfm"))
    np.testing.assert_allclose(imgs[0], imgs[1], rtol=2e-5, atol=1e-7)


def test_bench_rccl_step_with_one_rank():
    """bench.py's N-rank step with the REAL backend on this 1-GPU box: VPT_BENCH_FORCE_DIST runs it with a single rank -- nccl (= RCCL)
    process group, the context's RCCL communicator created from an id carried by torch.distributed, blue-noise reset + render +
    vpt_allreduce_accum enqueued on the context's stream without host fences, barrier + max-over-ranks timing.  What a second GPU
    would add is the traffic inside ncclAllReduce."""
    env = dict(os.environ, VPT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--width", "320", "--height", "180", "--spp", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and "per_frame" not in d and "other_configs" not in d
