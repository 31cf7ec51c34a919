#!/usr/bin/env python3
"""bench.py -- Msamples/s of the hot path (BASELINE.json metric) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|sun|c1|c3|c4|c5|c3-sun|c5-sun] [--spp N]
                    [--scaling weak|strong] [--no-other-configs] [--no-cpu-baseline] [--no-per-frame] [--no-parity]
    python bench.py --gpus N                 (launches its own N ranks through torch.distributed.run) -- or, equivalently,
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" = one render of the named workload (default: BASELINE config 2 = dragon.vdb, 1920x1080, 64 spp, procedural
sun + sky): `spp` iterations of `volume_rt_kernel` per rank (raygen + trace + tail/resolve kernels, inputs resident in
HBM).  Rank 0 prints TWO things (round 6: the round-5 line carried everything and grew past what the driver parses):
  BENCH_DETAIL {...}   one earlier stdout line (also gpurun_out/bench_detail.json): every block this run measured --
      roofline       of the headline's dominant kernel (the tracer), all labelled fractions, see `roofline_block`
      other_configs  BASELINE configs 3, 4, 5 on their synthetic stand-ins at spec size, 2 steps each, each with
                     its own parity evidence at that size (`parity`: oracle on a pixel lattice across two record chunks / layout A-B);
                     with N > 1: configs 4 and 5 (the ones BASELINE assigns to 8 GPUs) striped over the ranks, one all-reduce each
      per_frame      the literal drop-in call: one vpt_render (1 iteration) + device sync per frame, as main.cpp:1822-1829
      cpu_baseline   the reference's own kernel built for the host (oracle/_ref) or the oracle, on a bounded sample of the same
                     frame -- and the HIP image of the SAME iterations compared with it (`parity_rel_l2`)
      c1_cpu_single_thread, weak / strong, the sky caches' gates
  {"metric": ...}      the LAST stdout line, < 6000 bytes (`headline_line`): the contract's keys, a trimmed `roofline`,
                       `cpu_baseline`, and per other config its rate, fraction and parity figure
With N > 1 every rank renders its own iteration stripe and the accumulation buffers are combined with ONE RCCL
all-reduce below the C ABI (vpt_allreduce_accum) inside the timed region, with no host synchronisation inside a step.
  --scaling strong (default) the job's `spp` iterations are split over the ranks (spp/N each): fixed total work --
                   north_star's own sentence (one frame's sample batches over the GPUs, reduced once)
  --scaling weak   every rank renders `spp` iterations: N*spp samples per pixel per step
Both rates travel in the line (`weak`, `strong`) whichever is asked for; at N = 1 they are the same job.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (guides/MI355X_MICROARCH.md)
DEFAULT_SPP = {"c1": 64, "sun": 64, "c2": 64, "c3": 256, "c3-sun": 256, "c4": 128, "c5": 512, "c5-sun": 512}


def make_scene(pkg, cfg, W, H, spp, grid_scale, dev, local_rank):
    """-> (SceneDesc, workload string, data string, W, H)"""
    if cfg in ("c1", "sun", "c2"):
        sd = pkg.scene.dragon_scene(W, H, cfg)
        workload = "dragon.vdb %dx%dx%dspp, %s" % (W, H, spp, {
            "c2": "procedural sun+sky (BASELINE config 2)", "sun": "sun NEE only, sky_mult=0 (config 2 without the sky LUTs)",
            "c1": "one point light, no atmosphere (BASELINE config 1)"}[cfg])
        data = "synthetic camera/lights on the reference's dragon.vdb grid (committed fixture)"
    elif cfg in ("c3", "c3-sun"):
        sd = pkg.scene.fireball_scene(W, H, n=256, sky=cfg == "c3")
        workload = "synthetic fireball 256^3 (density + heat, blackbody LUT) %dx%dx%dspp, %s (BASELINE config 3 stand-in)" % (
            W, H, spp, "procedural sun+sky" if cfg == "c3" else "sun only")
        data = "synthetic (fireball.vdb is not shipped with the reference)"
    elif cfg == "c4":
        shape = tuple(int(round(x * grid_scale)) for x in (1216, 704, 1024))
        grid = pkg.scene.cloud_grid_torch(shape, device=dev)
        sd = pkg.scene.cloud_scene(W, H, env=(2048, 1024), integrator=1, device_grid=grid)
        workload = "synthetic cloud %dx%dx%d f32 (%.2f GB) + 2048x1024 HDRI, vol_integrator, %dx%dx%dspp (BASELINE config 4 stand-in)" % (
            shape[2], shape[1], shape[0], grid.numel() * 4 / 1e9, W, H, spp)
        data = "synthetic (the Disney cloud is not shipped with the reference)"
    elif cfg in ("c5", "c5-sun"):
        if W == 1920 and H == 1080:
            W, H = 3840, 2160
        sd = pkg.scene.instanced_scene(W, H, n=128, grid=10, aperture=2.0, sky=cfg == "c5")
        workload = "100 instances of a synthetic 128^3 coloured-smoke grid over the octree, DOF, %s, %dx%dx%dspp (BASELINE config 5 stand-in)" % (
            "procedural sun+sky" if cfg == "c5" else "sun only", W, H, spp)
        data = "synthetic (colored_smoke.vdb is not shipped with the reference)"
    else:
        raise SystemExit("unknown --config " + cfg)
    if cfg in ("c2", "c3", "c4", "c5"):
        pkg.atmosphere.attach_default_atmosphere(sd, device=local_rank)
    return sd, workload, data, W, H


def flips_block(lib, hb):
    """share of the ground table's check rays whose binary32 ground point the full path finds one step (0.5 m) above the ground: the reference's own
    ray-to-ray noise, gated apart from the table's deviation (csrc/vpt_tail.hip: sky_dir_table_rays_kernel)"""
    import ctypes as C
    f = (C.c_float * 4)()
    lib.vpt_test_get_dir_table_flips.argtypes = [C.c_void_p, C.POINTER(C.c_float * 4)]
    if lib.vpt_test_get_dir_table_flips(hb.ctx.h, C.byref(f)) != 0:
        return None
    return {"fraction_centre_variant": float(f[0]), "largest_fraction_all_variants": float(f[1]),
            "mean_cost_centre_variant": float(f[2]), "largest_mean_cost_all_variants": float(f[3]), "accepted_mean_cost": 3e-4}


def roofline_block(cfg, W, H, spp, cs, st, samples_per_step, step_s, integrator, lean=False, compact=False):
    """cs: stats of a counted pass, st: HIP-event times of the last timed step; lean: the tracer resolved its finished paths itself
    (24 bytes out per ray instead of a 64-byte path record, csrc/vpt_device.h ResolveParams::lean).

    `frac` IS BASELINE.md 3 / SURVEY 8d VERBATIM: B = 32 N_d + 128 N_c + 32 N_e + 88 bytes per pixel-sample with the REFERENCE-defined look-up
    counts (every instance of the leaf at every step, the colour at every step, the first walk twice -- what render_kernel.cu evaluates and the
    oracle counts), x samples per step / whole-step wall time / 8 TB/s.  `kernel` names the dominant kernel of the step.  (Not a bound on this
    implementation where the reference evaluates and discards: config 5's 13 colour look-ups per sample put it above 1.)
    Next to it, recomputable from this line and profiles/ of the same commit:
      frac_kernel_issued_fetches: bytes the tracer itself has to move -- `lookup_bytes` (the trilinear fetches it ISSUES: density for the
          instances whose domain holds the point, colour at real collisions, emission) + `record_stream_bytes` (its own ray / path records:
          4 + 64 B read -- 4 + 48 B with compact ray records -- and 64 B -- 24 B when `lean` -- written per traced ray; self-imposed traffic) / the tracer's HIP-event duration
      frac_step_issued_fetches: the issued look-up bytes + the 88-byte framebuffer term, x samples / whole-step time
      hbm_measured_frac: FETCH_SIZE x 2 + WRITE_SIZE of the tracer (rocprofv3 --pmc, profiles/traffic.json) / its duration"""
    n = float(max(1, cs.samples))
    nd, nc, ne = cs.density_lookups / n, cs.color_lookups / n, cs.emission_lookups / n
    fd, fc, fe = cs.density_fetches / n, cs.color_fetches / n, cs.emission_fetches / n
    counted_iters = max(1.0, n / float(W * H))
    traced = cs.queued_rays / float(W * H) / counted_iters
    b_lookup = 32.0 * fd + 128.0 * fc + 32.0 * fe
    # the tracer's own record stream per traced ray: 4 B queue entry + the ray record it reads (64 B; behind a closed lens 32 B compact + the 16-byte head: csrc/vpt_device.h
    # TraceParams::compact_rays) + what it writes (a 64-byte path record; 16 + 8 B when it resolves the sample itself)
    b_records = (4.0 + (48.0 if compact else 64.0) + (24.0 if lean else 64.0)) * traced
    b_kernel = b_lookup + b_records
    b_step = b_lookup + 88.0
    b_ref = 32.0 * nd + 128.0 * nc + 32.0 * ne + 88.0
    kernel_name = "vpt::trace_vol_kernel" if integrator else "vpt::trace_kernel"
    trace_s = st.trace_ms * 1e-3
    achieved_kernel = b_kernel * samples_per_step / trace_s / 1e9 if trace_s > 0 else 0.0
    achieved = b_ref * samples_per_step / step_s / 1e9
    # BASELINE.md's formula charges bytes for look-ups the reference evaluates and DISCARDS (config 5: 13 colour look-ups per sample, :1662 vs :1673):
    # where that puts the figure above the peak it is not a fraction of anything -- `frac` is null, `frac_void` says why, the raw ratio is kept under
    # a name that does not claim to be one, and the config's roofline figure is `frac_kernel_issued_fetches` (bytes this tracer must move / its time)
    void = achieved > HBM_PEAK_GBS
    r = {
        "bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 3) if not void else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5) if not void else None, "frac_void": bool(void), "traffic": None,
        "definition": "BASELINE.md 3 verbatim: (32 N_d + 128 N_c + 32 N_e + 88) B per pixel-sample with the reference-defined look-up counts x samples/s of the whole step",
        "frac_kernel_issued_fetches": round(achieved_kernel / HBM_PEAK_GBS, 5),
        "achieved_kernel_issued_fetches": round(achieved_kernel, 3),
        "frac_step_issued_fetches": round(b_step * samples_per_step / step_s / 1e9 / HBM_PEAK_GBS, 5),
        "hbm_measured_frac": None,
        "bytes_per_sample": {"survey_8d_reference_counts": round(b_ref, 2), "lookup_bytes": round(b_lookup, 2), "record_stream_bytes": round(b_records, 2),
                             "kernel_must_move": round(b_kernel, 2), "step_issued_fetches": round(b_step, 2)},
        "per_sample": {"density_fetches": round(fd, 4), "color_fetches": round(fc, 4), "emission_fetches": round(fe, 4),
                       "density_fetches_answered_by_zero_mask": round(cs.density_zero_skips / n, 4),
                       "density_lookups_reference": round(nd, 4), "color_lookups_reference": round(nc, 4), "emission_lookups_reference": round(ne, 4),
                       "tracking_steps": round(cs.tracking_steps / n, 4), "skip_steps": round(cs.skip_steps / n, 4), "rays_traced_fraction": round(traced, 4)},
        "raygen_ms_per_step": round(st.raygen_ms, 3), "trace_ms_per_step": round(st.trace_ms, 3),
        "tail_resolve_ms_per_step": round(st.tail_ms, 3),
        # the tracer's own rate: rays it walks per second (the headline counts every pixel-sample, most of which start no walk on config 2)
        "tracer_grays_per_s": round(traced * samples_per_step / trace_s / 1e9, 3) if trace_s > 0 else None,
        "resolved_samples": bool(lean),
        # one launch of the dominant kernel traces the chunk rule's iterations (vpt_render_batch: <= 64 iterations, <= 16 GiB of records)
        "samples_per_launch": W * H * max(1, min(64, spp, (16 << 30) // (W * H * 64))),
        "note": "kernel times: HIP events on the context's stream around every launch of the LAST timed step",
    }
    if void:
        r["reference_count_bytes_over_peak"] = round(achieved / HBM_PEAK_GBS, 3)
        r["frac_void_reason"] = ("BASELINE.md 3's bytes per sample count every look-up the reference evaluates, used or not (render_kernel.cu:1662 vs :1673): "
                                 "x samples/s they exceed the 8 TB/s peak, so the formula bounds nothing here; this config's roofline figure is frac_kernel_issued_fetches")
        r["frac_promoted"] = {"name": "frac_kernel_issued_fetches", "value": r["frac_kernel_issued_fetches"]}
    # measured HBM traffic of the dominant kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very
    # command, corrected per MI355X_MICROARCH.md; tools/profile_bench.sh + tools/make_traffic_json.py write the file)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f).get(cfg)
        if tj and tj.get("width") == W and tj.get("height") == H:
            per_sample = tj["bytes_per_launch"] / float(tj["samples_per_launch"])
            ipl = max(1, min(64, spp, (16 << 30) // (W * H * 64)))      # iterations per tracer launch: the chunk rule of vpt_render_batch
            launches = -(-spp // ipl)
            r["traffic"] = round(per_sample * samples_per_step / launches / 1e9, 4)
            r["traffic_unit"] = "GB per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/%s at commit %s)" % (tj.get("source", "?"), tj.get("commit", "?"))
            r["traffic_source"] = "profiles/" + tj.get("source", "?")
            r["traffic_commit"] = tj.get("commit")
            r["launches_per_step"] = launches
            if trace_s > 0:
                r["hbm_measured_frac"] = round(per_sample * samples_per_step / trace_s / 1e9 / HBM_PEAK_GBS, 5)
            # the bound that binds: vector-instruction issue at partial lane occupancy.  ONE number per kernel:
            #   useful_lane_issue = (fraction of the SIMDs' issue cycles spent on VALU instructions, priced at the kernel's
            #                        static instruction mix) x (active lanes per VALU instruction / 64)
            # from the SQ counters of the same rocprofv3 passes (profiles/<source>, taken at `commit`)
            kernels = {}
            for kname, vk in (tj.get("valu") or {}).items():
                if "valu_issue_busy_static_mix" in vk:
                    e = dict(vk)
                    e["useful_lane_issue"] = round(vk["valu_issue_busy_static_mix"] * vk["active_lanes_per_valu_instruction"] / 64.0, 3)
                    kernels[kname] = e
            if kernel_name in kernels:
                r["valu"] = {"bound": "VALU issue x lane occupancy", "kernel": kernel_name, "useful_lane_issue": kernels[kernel_name]["useful_lane_issue"],
                             "kernels": kernels, "source": "profiles/" + tj.get("source", "?"), "commit": tj.get("commit")}
    return r


def _kernel_commit():
    """the commit that last changed what the timed library is built from (csrc/, include/, build.py); None outside a git checkout (the GPU box's snapshot has no .git:
    there the stamp written by tools/stamp_commit.py travels in profiles/kernel_commit.txt)"""
    import subprocess
    try:
        r = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "volumetric-path-tracer_amd/csrc", "include", "volumetric-path-tracer_amd/build.py"],
                           capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            return r.stdout.strip()
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "profiles", "kernel_commit.txt")) as f:
            return f.read().split()[0]
    except Exception:
        return None


def headline_line(d):
    """The ONE final JSON line (round 6): the contract's keys, a trimmed `roofline`, `cpu_baseline`, and per other config only its rate, fraction and parity.
    Everything else this run measured is the DETAIL record (`emit`): an earlier stdout line and gpurun_out/bench_detail.json.  The round-5 line carried all
    of it and grew to 22 KB, which the driver no longer parsed."""
    rf = d["roofline"]
    r = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_void", "traffic")}
    r["definition"] = "(32 N_d + 128 N_c + 32 N_e + 88) B per pixel-sample (BASELINE.md 3, reference-defined look-up counts) x samples of one launch / whole-step time / 8 TB/s"
    r["bytes_per_sample"] = rf["bytes_per_sample"]["survey_8d_reference_counts"]
    r["samples_per_launch"] = rf.get("samples_per_launch")
    for k in ("frac_kernel_issued_fetches", "hbm_measured_frac", "tracer_grays_per_s", "cold_view_msamples_per_s", "launches_per_step"):
        r[k] = rf.get(k)
    r["kernel_ms"] = {"raygen": rf["raygen_ms_per_step"], "trace": rf["trace_ms_per_step"], "tail": rf["tail_resolve_ms_per_step"], "timer": "HIP events on the context's stream, last timed step"}
    valu = rf.get("valu") or {}
    r["useful_lane_issue"] = valu.get("useful_lane_issue")
    r["source"] = valu.get("source") or rf.get("traffic_source")
    r["commit"] = valu.get("commit") or rf.get("traffic_commit")
    if rf.get("frac_void"):
        r["frac_promoted"] = rf.get("frac_promoted")
    c = d["config"]
    out = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {k: c.get(k) for k in ("workload", "width", "height", "spp_per_gpu", "spp_job", "parallelism", "collective", "frames_in_flight")}
    out["roofline"] = r
    if d.get("cpu_baseline"):
        out["cpu_baseline"] = d["cpu_baseline"]
    for name in ("weak", "strong"):
        if d["n_gpus"] > 1 and name in d:
            out[name] = {k: d[name][k] for k in ("value", "ms_per_step", "spp_per_gpu", "spp_job", "frames_in_flight")}
    if d.get("per_frame"):
        pf = d["per_frame"]
        out["per_frame"] = {"value": pf["value"], "ms_per_frame": pf["ms_per_frame"], "frame_by_frame_value": (pf.get("frame_by_frame") or {}).get("value")}
    others = []
    for o in d.get("other_configs") or []:
        orf, par = o["roofline"], o.get("parity") or {}
        e = {"workload": o["config"]["workload"].split(" (BASELINE")[0][:96], "config": o.get("name"), "value": o["value"], "ms_per_step": o["ms_per_step"], "n_gpus": o.get("n_gpus", 1),
             "spp_per_gpu": o["config"]["spp_per_gpu"], "frac": orf.get("frac"), "traffic": orf.get("traffic"), "trace_ms": orf.get("trace_ms_per_step"),
             "parity_rel_l2": par.get("rel_l2"), "parity_kind": par.get("kind")}
        if orf.get("frac_void"):
            e["frac_promoted"] = orf["frac_promoted"]["value"]
        others.append(e)
    if others:
        out["other_configs"] = others
    c1 = d.get("c1_cpu_single_thread")
    if c1:
        out["c1_cpu_single_thread"] = {"cpu_msamples_per_s": c1["cpu"]["value"], "hip_msamples_per_s": c1["hip"]["value"], "parity_rel_l2": c1["parity_rel_l2"], "kind": c1["cpu"]["kind"]}
    out["kernel_commit"] = d.get("kernel_commit")
    out["detail"] = "stdout line 'BENCH_DETAIL {...}' before this one; gpurun_out/bench_detail.json"
    return out


def emit(d, detail_file):
    """rank 0: the detail record (a stdout line that does not start with '{', and a file); returns the headline, which main() prints as the LAST stdout line"""
    d["kernel_commit"] = _kernel_commit()
    detail = json.dumps(d)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(detail_file)), exist_ok=True)
        with open(detail_file, "w") as f:
            f.write(detail + "\n")
    except OSError:
        pass
    print("BENCH_DETAIL " + detail, flush=True)
    line = json.dumps(headline_line(d))
    if len(line) > 6000:                      # the driver's parser stopped at 22 KB; keep far below
        raise SystemExit("bench.py: the headline line grew to %d bytes (limit 6000)" % len(line))
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=0, help="iterations per step (weak: per rank; strong: of the whole job); default: the config's own")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="N > 1: strong (default, north_star's sentence: ONE job's iterations split over the ranks and reduced once) or weak (every rank renders the job's spp)")
    ap.add_argument("--grid-scale", type=float, default=1.0, help="c4: linear scale of the 1024x704x1216 cloud grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-per-frame", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the per-config parity leg of other_configs")
    ap.add_argument("--no-c4-oracle", action="store_true", help="config 4's parity leg: the layout A/B only (no 3.5 GB host copy, no oracle lattice)")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="independent frames a rank keeps in flight (contexts dealt round robin); 0 = 3 for strong-scaling "
                    "steps of <= 12 iterations per rank, else 1.  With N = 1 a value > 1 makes every step a fresh frame")
    ap.add_argument("--no-c1", action="store_true", help="skip the config-1 block (reference kernel on ONE host thread, 512x512x16spp)")
    ap.add_argument("--cpu-iters", type=int, default=32)
    ap.add_argument("--frames", type=int, default=64, help="frames of the per-frame (vpt_render + sync) measurement")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"), help="where the detail record is written besides stdout")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- the same command under torch.distributed.run, one rank per
        # GPU, rendezvous on 127.0.0.1 (the container's hostname may not resolve).  exec: rank 0's JSON line is this
        # process's stdout, the launcher's exit code is this command's.
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    import numpy as np
    import torch
    import __graft_entry__ as ge
    ge.build()
    pkg = ge.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start `python bench.py --gpus N` plainly (it launches its own ranks) or "
                         "under torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; VPT_BENCH_BACKEND=gloo lets several ranks share one GPU (functional check of the
    # N > 1 path on a 1-GPU box: RCCL refuses duplicate devices, gloo stages the all-reduce through the host)
    backend = os.environ.get("VPT_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    local_rank = dev_index
    import torch.distributed as dist
    # VPT_BENCH_FORCE_DIST=1: run the N-rank step (process group, RCCL communicator under the C ABI, stream-ordered reduce) with a
    # single rank -- the functional check of that code path on a 1-GPU box (tests/test_gpu_bench_ranks.py)
    multi = world > 1 or bool(os.environ.get("VPT_BENCH_FORCE_DIST"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = args.config
    job_spp = args.spp or DEFAULT_SPP.get(cfg, 64)

    def rank_spp(scaling, job):
        # strong: the job's iterations striped over the ranks: rank r renders iterations r, r + G, ... (one more on the first spp % G ranks);
        # the all-reduce weights every rank's mean with its own count.  weak: every rank renders the job's spp
        return len(range(rank, job, world)) if (scaling == "strong" and world > 1) else job

    def measure(cfg, job, steps, warmup, W, H, with_extras, scaling=None):
        """one workload (`job` = the config's iterations per step as BASELINE names them): returns the dict of the JSON line (rank 0) or None"""
        scaling = scaling or args.scaling
        spp = rank_spp(scaling, job)                                                    # iterations THIS rank renders per step
        job_iters = job * world if (world > 1 and scaling == "weak") else job           # iterations of the whole job per step
        sd, workload, data, W, H = make_scene(pkg, cfg, W, H, job, args.grid_scale, dev, local_rank)
        # scene set-up as a rank pays it at start (outside the timed region): texture uploads / adoption, the re-lay of big grids
        # into corner quads (config 4: 3.5 GB -> 14 GB on the GPU), the host octree and its candidate lists, buffer allocation
        torch.cuda.synchronize(dev)
        t_setup = time.perf_counter()
        hb = pkg.scene.HipBinding(sd, device=local_rank)
        hb.sync()
        torch.cuda.synchronize(dev)
        t_setup = time.perf_counter() - t_setup
        first_it, stride, bn_pre = pkg.dist.stripe(rank, world)
        bn0 = hb.blue_noise.clone()
        torch.cuda.synchronize(dev)
        use_comm = multi and backend == "nccl"
        # FRAMES IN FLIGHT (round 5): a strong-scaling step is ONE independent frame (a fresh render of this rank's stripe + the one all-reduce), and a
        # k-iteration launch of the persistent tracer pays ~0.25 ms of start-up and drain whatever k is -- 8 iterations per rank run at 68 % of
        # the batch rate.  Consecutive frames are independent, so a rank keeps F of them in flight: F contexts (own stream, own buffers, own
        # communicator), steps dealt round robin; frame f + 1's kernels run under frame f's drain and reduce.  Measured on one GPU
        # (profiles/r05_frames_in_flight.txt): 8 iterations per frame 1.093 -> 0.879 ms with F = 3; no gain from 16 iterations up, a loss at 64.
        in_flight = args.frames_in_flight or (3 if (world > 1 and scaling == "strong" and 0 < spp <= 12) else 1)
        fresh_frames = multi or in_flight > 1             # every step a fresh frame (else: the progressive render carries on)
        hbs = [hb] + [pkg.scene.HipBinding(sd, device=local_rank) for _ in range(in_flight - 1)]
        if use_comm:
            for h in hbs:
                pkg.dist.init_comm(h.ctx)                # RCCL communicator of this context (id carried by torch.distributed)
        # everything a step enqueues goes to its context's own stream; torch ops on it through an ExternalStream view
        cstreams = [torch.cuda.ExternalStream(h.ctx.stream, device=dev) for h in hbs]
        step_no = [0]
        last_reduce = [None]                              # event behind the previous frame's all-reduce (frames in flight: see below)

        def one_step():
            h = hbs[step_no[0] % in_flight]
            cs_ = cstreams[step_no[0] % in_flight]
            step_no[0] += 1
            if not fresh_frames:
                # the next `spp` iterations of the progressive render (iteration indices, blue-noise table and running means
                # carry on from the previous step, as consecutive frames of the reference do): no host round trip between steps
                h.render(spp)
                return
            # a step of the N-rank job: this rank's stripe of a fresh render, then the one all-reduce.  All of it is enqueued
            # on the context's stream in order -- the next step's kernels queue behind the reduce, no host fence in between.
            with torch.cuda.stream(cs_):
                h.blue_noise.copy_(bn0)
            if bn_pre:
                h.ctx.blue_noise_advance(h.blue_noise, bn_pre, sd.width * sd.height)
            if spp:                                         # (a rank beyond the job's iterations renders nothing and carries weight 0)
                h.render(spp, iter_stride=stride, iteration=first_it)
            if not multi:
                return
            if use_comm:
                # frames in flight reduce through their own communicators on their own streams: the collectives are chained (each starts behind the previous
                # frame's, on every rank alike), so that two communicators' kernels never wait for each other across ranks in different orders
                if in_flight > 1 and last_reduce[0] is not None:
                    cs_.wait_event(last_reduce[0])
                pkg.dist.combine_means(h.accum, spp, ctx=h.ctx)
                if in_flight > 1:
                    ev_ = torch.cuda.Event()
                    ev_.record(cs_)
                    last_reduce[0] = ev_
            else:
                h.sync()                                    # host-staged fallback (gloo): ctx stream -> torch's stream
                pkg.dist.combine_means(h.accum, spp)
                torch.cuda.synchronize(dev)

        def fence():
            if multi:
                dist.barrier()
            torch.cuda.synchronize(dev)

        hb.kp.iteration = first_it
        for _ in range(warmup):
            one_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_step()
        fence()
        elapsed = time.perf_counter() - t0
        st = hb.ctx.stats()                               # per-kernel HIP-event times of the LAST step (events on the ctx stream)
        if multi:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        samples_per_step_rank = W * H * spp
        value = W * H * job_iters * steps / elapsed / 1e6
        out = None
        if rank == 0:
            # ---- look-up counts: an untimed 2-iteration counted pass
            hb.ctx.set_counting(True)
            hb.blue_noise.copy_(bn0)
            torch.cuda.synchronize(dev)
            hb.render(2, iter_stride=stride, iteration=first_it)
            hb.sync()
            cs = hb.ctx.stats()
            hb.ctx.set_counting(False)
            step_s = elapsed / steps
            import ctypes as C
            lib = pkg.load_library()
            cstate = (C.c_int * 8)()
            lib.vpt_test_get_cache_state.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            lib.vpt_test_get_cache_state(hb.ctx.h, cstate)
            compact = float(sd.camera.lens_radius) == 0.0 and "VPT_NO_COMPACT_RAYS" not in os.environ and "VPT_NO_HEADS" not in os.environ
            roofline = roofline_block(cfg, W, H, spp, cs, st, samples_per_step_rank, step_s, sd.kp.integrator, lean=bool(cstate[6]), compact=compact)
            roofline["compact_ray_records"] = bool(compact)
            if not multi:
                # what the per-view caches of the environment tail cost to build (once per view, in the warm-up): one step right after
                # vpt_invalidate_sky_tables against the timed step
                hb.blue_noise.copy_(bn0)
                torch.cuda.synchronize(dev)
                hb.ctx.invalidate_sky_tables()
                tb = time.perf_counter()
                hb.render(spp, iteration=0)
                hb.sync()
                cold_s = time.perf_counter() - tb
                roofline["cache_build_ms_per_view"] = round(max(0.0, cold_s - step_s) * 1e3, 3)
                # the headline's timed steps render a still view whose per-view caches were built in the warm-up; a COLD frame (new view: caches
                # rebuilt inside the step) is this
                roofline["cold_view_ms_per_step"] = round(max(cold_s, step_s) * 1e3, 3)
                roofline["cold_view_msamples_per_s"] = round(W * H * spp / max(cold_s, step_s) / 1e6, 3)
                roofline["caches_in_use"] = dict(zip(("sky_patch", "never_traced", "sky_dome", "dome_variants", "cam_table", "dir_table", "resolved_samples"), [int(x) for x in list(cstate)[:7]]))
            out = {
                "metric": "Msamples/s (W*H*spp/s)", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
                "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
                "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": data,
                "config": {"workload": workload, "width": W, "height": H, "spp_per_gpu": spp, "spp_job": job_iters,
                           "parallelism": "iteration-striped x%d + 1 RCCL all-reduce under the C ABI" % world if world > 1 else "1 GPU",
                           # what carried the reduce: the rank count read back from the context's RCCL communicator, or the
                           # host-staged torch.distributed fallback (VPT_BENCH_BACKEND=gloo: ranks sharing one GPU)
                           "collective": ({"backend": "rccl (vpt_allreduce_accum)", "comm_ranks": int(hb.ctx.comm_nranks)} if use_comm else
                                          {"backend": backend + " (torch.distributed, host-staged)", "comm_ranks": world}) if multi else None,
                           "frames_in_flight": in_flight,
                           "arithmetic": "strict (no FMA contraction, fixed-sequence log/sin/cos)",
                           "scene_setup_s": round(t_setup, 3)},
                "roofline": roofline,
            }
            # per-frame sky tables of the environment tail: were ground tables in use, and their self-measured error
            try:
                import ctypes as C
                lib = pkg.load_library()
                lib.vpt_test_get_dir_table_error.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_uint)]
                built, err, cell = C.c_int(0), C.c_float(0), C.c_uint(0)
                if lib.vpt_test_get_dir_table_error(hb.ctx.h, C.byref(built), C.byref(err), C.byref(cell)) == 0:
                    chk = (C.c_float * 8)()
                    lib.vpt_test_get_dir_table_check.argtypes = [C.c_void_p, C.POINTER(C.c_float * 8)]
                    lib.vpt_test_get_dir_table_check(hb.ctx.h, C.byref(chk))
                    out["config"]["sky_ground_table"] = {
                        "built": bool(built.value), "in_use": bool(chk[4]),
                        "interpolation_error_at_cell_centres": float(err.value), "interpolation_error_accepted_below": 5e-4,
                        "variants_in_use": int(chk[5]),
                        "vs_full_path_along_real_rays": {"rays_centre_variant": int(chk[2]), "max_relative_difference": float(chk[1]), "accepted_below": 2e-2,
                                                         "fraction_unflipped_above_1e-3": (float(chk[3]) / float(chk[2])) if chk[2] else None, "accepted_fraction": 0.005,
                                                         "all_variants_max": float(chk[6]), "all_variants_largest_fraction_unflipped_above_1e-3": float(chk[7]),
                                                         "flipped_rays": flips_block(lib, hb)}}
                pa, pb = C.c_ulonglong(0), C.c_ulonglong(0)
                lib.vpt_test_get_sky_patch_coverage.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
                if lib.vpt_test_get_sky_patch_coverage(hb.ctx.h, C.byref(pa), C.byref(pb)) == 0:
                    # per-pixel sky patches: pixels whose untraced samples take their sky value from a checked bilinear patch
                    out["config"]["sky_patch"] = {"pixels": int(pa.value), "with_patch": int(pb.value), "centre_check_relative": 1e-3}
            except Exception as e:                                    # reporting only
                out["config"]["sky_ground_table"] = {"error": str(e)}
            if not with_extras and not multi and not args.no_parity:
                out["parity"] = config_parity(cfg, hb, sd, bn0, W, H, spp)
            if with_extras and not multi and not args.no_per_frame:
                out["per_frame"] = per_frame(hb, sd, bn0, W, H)
            if with_extras and not multi and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(hb, sd, bn0, W, H, spp)
        if multi:
            dist.barrier()
        for h in hbs:
            h.ctx.close()
        del hb, hbs
        torch.cuda.empty_cache()
        return out

    def per_frame(hb, sd, bn0, W, H):
        """the literal drop-in call (main.cpp:1822-1829): one launch per iteration, device sync after every frame"""
        hb.blue_noise.copy_(bn0)
        torch.cuda.synchronize(dev)
        hb.kp.iteration = 0
        for _ in range(4):
            hb.render_frame()
            hb.sync()
        tf = time.perf_counter()
        for _ in range(args.frames):
            hb.render_frame()
            hb.sync()
        dt = time.perf_counter() - tf
        st = hb.ctx.stats()
        out = {"value": round(W * H * args.frames / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_frame": round(dt / args.frames * 1e3, 4),
               "frames": args.frames, "kernels_ms_last_frame": {"raygen": round(st.raygen_ms, 4), "trace": round(st.trace_ms, 4), "tail_resolve": round(st.tail_ms, 4)},
               "call": "vpt_render (iter_count 1) + vpt_sync per frame, as source/main.cpp:1822-1829",
               "frame_ahead": "on (the product's default): from the second identical call on a call traces the rays of the next 2..16 iterations in one launch, the following calls "
                              "run only their tail; every buffer after every frame bit-identical to frame by frame (tests/test_gpu_edge.py)"}
        # ... and the same loop with one launch per frame (VPT_NO_FRAME_AHEAD=1, as rounds 1-4): a second context, the switch is read when a context is created
        if "VPT_NO_FRAME_AHEAD" not in os.environ:
            os.environ["VPT_NO_FRAME_AHEAD"] = "1"
            try:
                h2 = pkg.scene.HipBinding(sd, device=local_rank)
            finally:
                del os.environ["VPT_NO_FRAME_AHEAD"]
            for _ in range(4):
                h2.render_frame(); h2.sync()
            tf = time.perf_counter()
            for _ in range(args.frames):
                h2.render_frame(); h2.sync()
            dt2 = time.perf_counter() - tf
            s2 = h2.ctx.stats()
            h2.ctx.close()
            out["frame_by_frame"] = {"value": round(W * H * args.frames / dt2 / 1e6, 3), "ms_per_frame": round(dt2 / args.frames * 1e3, 4),
                                     "kernels_ms_last_frame": {"raygen": round(s2.raygen_ms, 4), "trace": round(s2.trace_ms, 4), "tail_resolve": round(s2.tail_ms, 4)}}
        return out

    def config_parity(cfg, hb, sd, bn0, W, H, spp):
        """parity evidence of configs 3-5 AT THE SIZE THAT IS TIMED, carried in the bench line:
        device-side grids (config 4): the corner-quad layout against a counted pass of the dense layout -- same counts, same bits;
        host grids (configs 3, 5): two record chunks (a launch boundary at full size) against the oracle on a lattice of the frame"""
        ipl = max(1, min(64, spp, (16 << 30) // (W * H * 64)))
        host_grids = all(isinstance(v[1], np.ndarray) for v in sd.volumes)
        counts = ("samples", "density_lookups", "color_lookups", "emission_lookups", "tracking_steps", "skip_steps", "queued_rays", "density_fetches")
        if not host_grids:
            def counted(h):
                h.ctx.set_counting(True)
                h.blue_noise.copy_(bn0)
                torch.cuda.synchronize(dev)
                h.render(2, iteration=0)
                h.sync()
                return h.ctx.stats()
            sa = counted(hb)
            os.environ["VPT_GRID_LAYOUT"] = "dense"
            try:
                hd = pkg.scene.HipBinding(sd, device=local_rank)
            finally:
                del os.environ["VPT_GRID_LAYOUT"]
            sb = counted(hd)
            same_counts = all(getattr(sa, c) == getattr(sb, c) for c in counts)
            same_bits = bool(torch.equal(hb.accum, hd.accum) and torch.equal(hb.depth, hd.depth))
            hd.ctx.close()
            del hd
            torch.cuda.empty_cache()
            hb.ctx.set_counting(False)
            layout = {"iterations": 2, "counts_equal_dense_layout": same_counts, "buffers_bit_identical_dense_layout": same_bits,
                      "note": "re-laid (corner-quad) density grid vs VPT_GRID_LAYOUT=dense on the same device grid"}
            # ... and against the ORACLE at this size like configs 3 and 5 (round 5): a host copy of the device grid (3.5 GB at spec), two record chunks on a
            # lattice of the frame.  --no-c4-oracle keeps the layout comparison only.
            if args.no_c4_oracle:
                return dict(layout, kind="layout")
            import copy
            sd_host = copy.copy(sd)
            sd_host.volumes = [(v[0], v[1].detach().cpu().numpy() if not isinstance(v[1], np.ndarray) else v[1], v[2], v[3]) for v in sd.volumes]
            res = oracle_lattice(cfg, hb, sd_host, bn0, W, H, ipl)
            res["layout"] = layout
            return res
        return oracle_lattice(cfg, hb, sd, bn0, W, H, ipl)

    def oracle_lattice(cfg, hb, sd, bn0, W, H, ipl):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding
        step = {"c3": 17, "c5": 131, "c4": 67}.get(cfg, 17)
        iters = 2 * ipl if cfg != "c4" else ipl + 8      # two record chunks either way (a launch boundary); config 4's vol_integrator walks cost the CPU ~5x per sample
        cores = os.cpu_count() or 1
        ob = oracle_binding.OracleBinding(sd)
        tc = time.perf_counter()
        ob.render(iters, nthreads=cores, pixel_step=step)
        dtc = time.perf_counter() - tc
        hb.blue_noise.copy_(bn0)
        torch.cuda.synchronize(dev)
        hb.render(iters, iteration=0)
        hb.sync()
        got = hb.accum.cpu().numpy()[::step].astype(np.float64)
        ref = ob.accum[::step].astype(np.float64)
        rel = float(np.sqrt(((got - ref) ** 2).sum()) / max(1e-30, np.sqrt((ref ** 2).sum())))
        dgot, dref = hb.depth.cpu().numpy()[::step], ob.depth[::step]
        return {"kind": "oracle", "iterations": iters, "record_chunks": 2, "pixel_step": step, "pixels": int(got.shape[0]), "rel_l2": rel,
                "depth_pixels_differing": int((dgot != dref).sum()), "tolerance": 1e-3, "cpu_seconds": round(dtc, 1),
                "note": "HIP accum / depth after %d iterations (2 record chunks) vs the oracle on every %dth pixel of the %dx%d frame" % (iters, step, W, H)}

    def c1_single_thread():
        """BASELINE config 1 as SURVEY 8d specifies it: dragon.vdb, 512 x 512, 16 spp, ONE point light, no atmosphere -- the reference's own kernel
        (oracle/_ref; the oracle where that library is absent: same image bit for bit) on ONE host thread, the whole config, next to the HIP path on the
        very same 16 iterations and the parity of the two."""
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding
        import ref_binding
        Wc, Hc, sppc = 512, 512, 16
        sd1 = pkg.scene.dragon_scene(Wc, Hc, "c1")
        # "no atmosphere" = sun_mult = sky_mult = 0: the reference's kernel still CALLS sample_atmosphere (:1840) and multiplies by zero, so the
        # compiled reference needs finite tables (without them NaN x 0 trips its NaN guard and the image is black); the HIP path gets the same tables
        pkg.atmosphere.attach_default_atmosphere(sd1, device=local_rank)
        use_ref = ref_binding.have_ref()
        ob = ref_binding.RefBinding(sd1) if use_ref else oracle_binding.OracleBinding(sd1)
        tc = time.perf_counter()
        ob.render(sppc, nthreads=1)
        dtc = time.perf_counter() - tc
        h1 = pkg.scene.HipBinding(sd1, device=local_rank)
        h1.render(sppc, iteration=0)
        h1.sync()
        got, ref = h1.accum.cpu().numpy().astype(np.float64), ob.accum.astype(np.float64)
        # the point light's falloff is 1 / length(lp * lp - pp * pp) (light.h:104-121 as written): a handful of samples land where that vanishes and
        # come out at 1e30 and more on BOTH sides -- an image-wide L2 is then the last bits of those fireflies.  The figure is taken over the pixels
        # below the 99.9th percentile of the reference's brightness; the rest are counted and compared RELATIVELY, pixel by pixel.
        lum = ref.max(1)
        keep = lum <= np.quantile(lum, 0.999)
        rel = float(np.sqrt(((got[keep] - ref[keep]) ** 2).sum()) / max(1e-30, np.sqrt((ref[keep] ** 2).sum())))
        ff = ~keep
        ff_rel = float((np.abs(got[ff] - ref[ff]).max(1) / np.maximum(lum[ff], 1e-30)).max()) if ff.any() else 0.0
        ddiff = int((h1.depth.cpu().numpy() != ob.depth).sum())
        bn1 = h1.blue_noise.clone()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(20):
            h1.render(sppc)
        h1.sync()
        dth = (time.perf_counter() - t1) / 20
        h1.ctx.close()
        return {"workload": "dragon.vdb %dx%dx%dspp, one point light, no atmosphere (BASELINE config 1)" % (Wc, Hc, sppc),
                "cpu": {"value": round(Wc * Hc * sppc / dtc / 1e6, 4), "unit": "Msamples/s", "cores": 1, "seconds": round(dtc, 2),
                        "kind": "reference" if use_ref else "port",
                        "what": "the reference's render_kernel.cu built for the host (oracle/_ref), 1 thread, all 16 iterations" if use_ref else "oracle, 1 thread, all 16 iterations"},
                "hip": {"value": round(Wc * Hc * sppc / dth / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dth * 1e3, 4)},
                "parity_rel_l2": rel, "parity_depth_pixels_differing": ddiff, "tolerance": 1e-3,
                "parity_note": "relative L2 over the pixels below the 99.9th percentile of the reference's brightness; the %d brighter ones (fireflies of the point light's "
                               "falloff, up to %.1e) agree to %.1e of their own value" % (int(ff.sum()), float(lum.max()), ff_rel)}

    def cpu_baseline(hb, sd, bn0, W, H, spp):
        host_grids = all(isinstance(v[1], np.ndarray) for v in sd.volumes)
        if not host_grids:
            return None
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding
        import ref_binding
        cores = os.cpu_count() or 1
        # the reference's own kernel source compiled for the CPU (oracle/_ref/libvptref.so, shipped prebuilt) when it
        # is there, else the oracle restatement; both produce the same image bit for bit (tests/test_oracle_vs_ref.py)
        use_ref = ref_binding.have_ref()
        ob = ref_binding.RefBinding(sd) if use_ref else oracle_binding.OracleBinding(sd)
        tc = time.perf_counter()
        ob.render(args.cpu_iters, nthreads=cores)
        dtc = time.perf_counter() - tc
        # the SAME iterations with the HIP path, into the same (reset) buffers: full-size parity of the headline frame
        hb.blue_noise.copy_(bn0)
        torch.cuda.synchronize(dev)
        hb.render(args.cpu_iters, iteration=0)
        hb.sync()
        got, ref = hb.accum.cpu().numpy(), ob.accum
        rel = float(np.sqrt(((got.astype(np.float64) - ref) ** 2).sum()) / max(1e-30, np.sqrt((ref.astype(np.float64) ** 2).sum())))
        dgot, dref = hb.depth.cpu().numpy(), ob.depth
        what = ("reference render_kernel.cu built for the host (oracle/_ref, %d threads over pixels)" % cores) if use_ref \
            else "oracle (OpenMP over rows)"
        return {"value": round(W * H * args.cpu_iters / dtc / 1e6, 4), "unit": "Msamples/s", "cores": cores,
                "kind": "reference" if use_ref else "port",
                "sample": "%d of %d iterations of the same %dx%d frame, %s, %.1f s" % (args.cpu_iters, spp, W, H, what, dtc),
                "parity_rel_l2": rel, "parity_depth_max_abs_diff": float(np.abs(dgot - dref).max()),
                "parity_depth_pixels_differing": int((dgot != dref).sum()),
                "parity_note": "HIP accum / depth buffers vs this CPU render after the same %d iterations at full size (tolerance 1e-3 rel. L2)" % args.cpu_iters}

    out = measure(cfg, job_spp, args.steps, args.warmup, args.width, args.height, True)
    # BOTH scalings travel in the one line (round 5): `value` / `scaling` are the mode asked for (weak by default: what the driver's N = 1, 2, 4, 8
    # runs compare), `weak` and `strong` hold the job's rate either way -- strong is north_star's own sentence: ONE frame's sample batches split over
    # the GPUs and reduced once.  At N = 1 the two are the same job.
    other = "strong" if args.scaling == "weak" else "weak"
    if world > 1:
        o2 = measure(cfg, job_spp, args.steps, args.warmup, args.width, args.height, False, scaling=other)
    else:
        o2 = out
    if rank == 0 and out is not None and o2 is not None:
        for name, o in ((args.scaling, out), (other, o2)):
            out[name] = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"],
                         "spp_per_gpu": o["config"]["spp_per_gpu"], "spp_job": o["config"]["spp_job"], "n_gpus": world,
                         "frames_in_flight": o["config"]["frames_in_flight"]}
    # The other BASELINE configs at spec size.  N = 1: configs 3, 4, 5 on the one GPU.  N > 1 (round 6): the two configs BASELINE assigns to the 8-GPU node --
    # config 4 (128 spp: 16 per rank at N = 8) and config 5 (512 spp: 64 per rank) -- iteration-striped over the ranks like the headline, each step ending in its
    # ONE all-reduce (24.9 MB at 1080p, 99.5 MB at 4K) inside the timed region.  A launcher started by hand with VPT_BENCH_FORCE_DIST runs the headline only.
    if not args.no_other_configs and cfg == "c2" and (not multi or world > 1):
        others = []
        for oc in (("c3", "c4", "c5") if world == 1 else ("c4", "c5")):
            o = measure(oc, DEFAULT_SPP[oc], 2, 1, args.width, args.height, False, scaling="strong")
            if o:
                others.append({"name": oc, "config": o["config"], "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"],
                               "warmup": o["warmup"], "n_gpus": world, "scaling": o["scaling"], "data": o["data"], "roofline": o["roofline"], "parity": o.get("parity")})
        if out is not None:
            out["other_configs"] = others
    if not multi and rank == 0 and out is not None and cfg == "c2" and not args.no_c1 and not args.no_cpu_baseline:
        out["c1_cpu_single_thread"] = c1_single_thread()
    headline = emit(out, args.detail_file) if (rank == 0 and out is not None) else None
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    # the headline is the LAST thing this process writes to stdout: RCCL prints its version banner through C stdio, which a pipe buffers until the
    # process exits -- flushed here, ahead of the line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if headline is not None:
        print(headline, flush=True)


if __name__ == "__main__":
    main()
