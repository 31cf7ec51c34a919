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


def _lines(stdout):
    """-> (headline dict = the LAST stdout line, detail dict = the 'BENCH_DETAIL {...}' line before it)"""
    out = [l for l in stdout.splitlines() if l.strip()]
    heads = [l for l in out if l.startswith("{")]
    assert len(heads) == 1 and out[-1] == heads[0], "ONE JSON line, and it is the last line of stdout"
    assert len(heads[0]) < 6000, "the headline line must stay far below what the driver parses (%d bytes)" % len(heads[0])
    det = [l for l in out if l.startswith("BENCH_DETAIL ")]
    assert len(det) == 1
    return json.loads(heads[0]), json.loads(det[0][len("BENCH_DETAIL "):])


def test_two_ranks_on_one_gpu():
    env = dict(os.environ, VPT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--detail-file", os.devnull, "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--width", "320", "--height", "180", "--spp", "4", "--no-cpu-baseline", "--no-other-configs"]
    r = subprocess.run(cmd + ["--scaling", "weak"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d, det = _lines(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["unit"] == "Msamples/s"
    assert d["value"] > 0 and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["spp_per_gpu"] == 4
    # both scalings travel in the one line: the job's rate with every rank rendering the 4 iterations, and with the 4 split over the 2 ranks
    assert d["weak"]["value"] == d["value"] and d["weak"]["spp_per_gpu"] == 4 and d["weak"]["spp_job"] == 8
    assert d["strong"]["value"] > 0 and d["strong"]["spp_per_gpu"] == 2 and d["strong"]["spp_job"] == 4 and det["strong"]["n_gpus"] == 2
    # a strong-scaling step of a few iterations per rank is one independent frame: three of them in flight per rank (three contexts dealt round robin)
    assert d["strong"]["frames_in_flight"] == 3 and d["weak"]["frames_in_flight"] == 1
    # the DEFAULT with N > 1 (round 6) is north_star's own sentence -- fixed total work: the job's 4 iterations split over the 2 ranks
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d, det = _lines(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["spp_per_gpu"] == 2 and d["config"]["spp_job"] == 4 and d["value"] > 0
    assert d["strong"]["value"] == d["value"] and d["weak"]["spp_per_gpu"] == 4 and d["weak"]["value"] > 0


def test_eight_ranks_strong_scaling_dry_run():
    """`bench.py --gpus 8 --scaling strong` end to end with the job's 12 iterations NOT a multiple of the ranks (2 2 2 2 1 1 1 1), eight ranks
    sharing this box's one GPU through the gloo fallback: the striping, the weighted reduce and the bench line of the 8-rank job the driver
    runs on an 8-GPU node (there with one RCCL all-reduce under the C ABI)."""
    env = dict(os.environ, VPT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--detail-file", os.devnull, "--gpus", "8", "--steps", "1", "--warmup", "1",
           "--width", "256", "--height", "144", "--spp", "12", "--no-cpu-baseline", "--grid-scale", "0.0625"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d, det = _lines(r.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["spp_per_gpu"] == 2 and d["config"]["spp_job"] == 12          # rank 0 renders iterations 0 and 8
    assert abs(d["value"] - 256 * 144 * 12 / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["value"]
    # the two configs BASELINE assigns to the 8-GPU node travel with it, striped the same way (round 6): config 4's 128 iterations as 16 per rank,
    # config 5's 512 as 64 per rank, each step ending in its one all-reduce
    oc = {o["config"]: o for o in d["other_configs"]}
    assert sorted(oc) == ["c4", "c5"]
    assert oc["c4"]["spp_per_gpu"] == 16 and oc["c5"]["spp_per_gpu"] == 64 and oc["c4"]["n_gpus"] == 8
    for name, job in (("c4", 128), ("c5", 512)):
        o = [x for x in det["other_configs"] if x["name"] == name][0]
        assert o["scaling"] == "strong" and o["config"]["spp_job"] == job and o["value"] > 0
        assert abs(o["value"] - o["config"]["width"] * o["config"]["height"] * job / (o["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * o["value"]


def test_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what a scaling run issues): bench.py re-executes itself under
    torch.distributed.run, rank 0 prints ONE JSON line with n_gpus = 2"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["VPT_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--detail-file", os.devnull, "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--width", "320", "--height", "180", "--spp", "4", "--no-other-configs"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d, det = _lines(r.stdout)
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
    """the N = 1 output: the LAST stdout line is the headline (every key of the contract, a trimmed roofline, the CPU baseline), under 6000 bytes;
    everything else travels in the BENCH_DETAIL line before it"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--detail-file", os.devnull, "--steps", "2", "--warmup", "1", "--width", "320", "--height", "180",
           "--spp", "4", "--cpu-iters", "2", "--frames", "4", "--grid-scale", "0.125"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    h, d = _lines(r.stdout)
    # ---- the headline line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in h, k
    assert h["n_gpus"] == 1 and h["steps"] == 2 and h["warmup"] == 1 and h["higher_is_better"] is True
    assert h["vs_baseline"] is None and h["dtype"] == "f32" and "workload" in h["config"] and "model" not in h["config"]
    hr = h["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "definition", "bytes_per_sample", "samples_per_launch",
              "frac_kernel_issued_fetches", "hbm_measured_frac", "kernel_ms", "useful_lane_issue", "cold_view_msamples_per_s", "commit"):
        assert k in hr, k
    assert hr["bound"] == "hbm" and hr["unit"] == "GB/s" and hr["peak"] == 8000.0 and hr["kernel"] == "vpt::trace_kernel"
    assert abs(hr["frac"] - hr["achieved"] / hr["peak"]) < 1e-4 and 0 < hr["frac"] < 1 and hr["frac_void"] is False
    # achieved = algorithmic bytes per unit x units / time: recomputable from the line itself
    assert abs(hr["achieved"] - hr["bytes_per_sample"] * h["value"] * 1e6 / 1e9) <= 2e-3 * hr["achieved"]
    assert hr["samples_per_launch"] == 320 * 180 * 4
    cb = h["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == h["unit"]
    # full-frame parity of the same iterations: HIP vs the CPU render (the reference's kernel where oracle/_ref exists)
    assert cb["parity_rel_l2"] <= 1e-3 and cb["parity_depth_pixels_differing"] == 0
    assert h["per_frame"]["value"] > 0 and h["per_frame"]["frame_by_frame_value"] > 0
    assert [o["config"] for o in h["other_configs"]] == ["c3", "c4", "c5"]
    for o in h["other_configs"]:
        assert o["value"] > 0 and o["parity_rel_l2"] is not None and o["parity_rel_l2"] <= 1e-3
        assert (o["frac"] is not None and 0 < o["frac"] < 1) or 0 < o["frac_promoted"] < 1
    c1 = h["c1_cpu_single_thread"]
    assert c1["cpu_msamples_per_s"] > 0 and c1["hip_msamples_per_s"] > 0 and c1["parity_rel_l2"] <= 1e-3
    assert "weak" not in h and "strong" not in h              # (one GPU: the same job either way)
    # the commit stamped on the profile-derived figures is the commit that last changed the timed library's sources (where git can say)
    if h["kernel_commit"] and hr["commit"] and os.path.isdir(os.path.join(ROOT, ".git")):
        assert hr["commit"] == h["kernel_commit"], "profiles/traffic.json was taken at %s, the library is built from %s" % (hr["commit"], h["kernel_commit"])
    # ---- the detail record
    assert d["value"] == h["value"] and d["roofline"]["frac"] == hr["frac"]
    rf = d["roofline"]
    # `frac` is BASELINE.md 3's figure (reference-defined look-up counts, whole step); the tracer's own figures travel next to it
    for k in ("frac_kernel_issued_fetches", "frac_step_issued_fetches", "hbm_measured_frac", "tracer_grays_per_s", "cache_build_ms_per_view", "definition"):
        assert k in rf, k
    bs = rf["bytes_per_sample"]
    assert abs(bs["kernel_must_move"] - bs["lookup_bytes"] - bs["record_stream_bytes"]) < 0.02
    assert 0 < rf["frac_kernel_issued_fetches"] < 1 and 0 < rf["frac_step_issued_fetches"] < 1
    pf = d["per_frame"]
    assert pf["value"] > 0 and pf["frames"] == 4 and pf["ms_per_frame"] > 0
    oc = d["other_configs"]
    assert len(oc) == 3
    for o in oc:
        rfo = o["roofline"]
        assert o["value"] > 0 and "workload" in o["config"] and 0 < rfo["frac_step_issued_fetches"] < 1
        # no figure is above 1 under the name of a fraction: where BASELINE.md's bytes per sample x samples/s exceed the peak -- config 5: the
        # reference-defined counts charge the colour look-ups the reference evaluates and discards -- `frac` is null, `frac_void` says so, the raw ratio travels under
        # another name and the kernel's own figure is promoted
        if rfo["frac_void"]:
            assert rfo["frac"] is None and rfo["achieved"] is None and rfo["reference_count_bytes_over_peak"] > 1.0
            assert rfo["frac_promoted"]["name"] == "frac_kernel_issued_fetches" and 0 < rfo["frac_promoted"]["value"] < 1
        else:
            assert 0 < rfo["frac"] < 1
    assert d["weak"]["value"] == d["value"] == d["strong"]["value"] and d["weak"]["frames_in_flight"] == 1
    assert 0 < rf["cold_view_msamples_per_s"] <= d["value"] * 1.001
    assert pf["frame_by_frame"]["value"] > 0 and "frame_ahead" in pf
    c1 = d["c1_cpu_single_thread"]
    assert c1["cpu"]["cores"] == 1 and c1["cpu"]["value"] > 0 and c1["hip"]["value"] > 0 and c1["parity_rel_l2"] <= 1e-3 and c1["parity_depth_pixels_differing"] == 0


def test_frames_in_flight_through_rccl_single_rank():
    """the strong-scaling step with THREE frames in flight (three contexts, three RCCL communicators, the all-reduces chained by events) -- the N-rank code path with one
    rank (VPT_BENCH_FORCE_DIST=1: process group + communicators under the C ABI + stream-ordered reduces), which is what a 1-GPU box can run of it"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(VPT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--detail-file", os.devnull, "--steps", "7", "--warmup", "2", "--width", "320", "--height", "180", "--spp", "4",
           "--scaling", "strong", "--frames-in-flight", "3", "--no-cpu-baseline", "--no-other-configs", "--no-per-frame"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d, det = _lines(r.stdout)
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
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--detail-file", os.devnull, "--gpus", "1", "--steps", "3", "--warmup", "1", "--width", "320", "--height", "180", "--spp", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d, det = _lines(r.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and "per_frame" not in d and "other_configs" not in d
