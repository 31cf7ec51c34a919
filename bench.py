#!/usr/bin/env python3
"""bench.py -- Msamples/s of the hot path (BASELINE.json metric) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|sun|c1|c3|c4|c5] [--spp N]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" = one render of the named workload (default: BASELINE config 2 = dragon.vdb, 1920x1080,
64 spp, procedural sun + sky): `spp` iterations of `volume_rt_kernel` per rank (raygen + trace +
tail/resolve kernels, inputs resident in HBM).  The other BASELINE configs (c3 fireball emission,
c4 large cloud + HDRI + vol_integrator, c5 100 instances at 4K with DOF) run on their synthetic
stand-ins; they are reported in DESIGN.md, the driver's bench line is c2.  With N > 1
every rank renders its own iteration stripe (weak scaling: N*spp samples per pixel per
step) and the accumulation buffers are combined with ONE all-reduce (RCCL) inside the
timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (guides/MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=0, help="iterations per rank and step (default: the config's own)")
    ap.add_argument("--grid-scale", type=float, default=1.0, help="c4: linear scale of the 1024x704x1216 cloud grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=32)
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as ge
    ge.build()
    pkg = ge.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    W, H = args.width, args.height
    cfg = args.config
    spp = args.spp or {"c3": 256, "c4": 128, "c5": 512}.get(cfg, 64)
    if cfg in ("c1", "sun", "c2"):
        sd = pkg.scene.dragon_scene(W, H, cfg)
        workload = "dragon.vdb %dx%dx%dspp, %s" % (W, H, spp, {
            "c2": "procedural sun+sky (BASELINE config 2)", "sun": "sun NEE only, sky_mult=0 (config 2 without the sky LUTs)",
            "c1": "one point light, no atmosphere (BASELINE config 1)"}[cfg])
        data = "synthetic camera/lights on the reference's dragon.vdb grid (committed fixture)"
    elif cfg == "c3":
        sd = pkg.scene.fireball_scene(W, H, n=256)
        workload = "synthetic fireball 256^3 (density + heat, blackbody LUT) %dx%dx%dspp, sun (BASELINE config 3 stand-in)" % (W, H, spp)
        data = "synthetic (fireball.vdb is not shipped with the reference)"
    elif cfg == "c4":
        shape = tuple(int(round(x * args.grid_scale)) for x in (1216, 704, 1024))
        grid = pkg.scene.cloud_grid_torch(shape, device=dev)
        sd = pkg.scene.cloud_scene(W, H, env=(2048, 1024), integrator=1, device_grid=grid)
        workload = "synthetic cloud %dx%dx%d f32 (%.2f GB) + 2048x1024 HDRI, vol_integrator, %dx%dx%dspp (BASELINE config 4 stand-in)" % (
            shape[2], shape[1], shape[0], grid.numel() * 4 / 1e9, W, H, spp)
        data = "synthetic (the Disney cloud is not shipped with the reference)"
    elif cfg == "c5":
        if args.width == 1920 and args.height == 1080:
            W, H = 3840, 2160
        sd = pkg.scene.instanced_scene(W, H, n=128, grid=10, aperture=2.0)
        workload = "100 instances of a synthetic 128^3 coloured-smoke grid over the octree, DOF, %dx%dx%dspp (BASELINE config 5 stand-in)" % (W, H, spp)
        data = "synthetic (colored_smoke.vdb is not shipped with the reference)"
    else:
        raise SystemExit("unknown --config " + cfg)
    if cfg in ("c2", "c4"):
        pkg.atmosphere.attach_default_atmosphere(sd, device=local_rank)
    hb = pkg.scene.HipBinding(sd, device=local_rank)
    first_it, stride, bn_pre = pkg.dist.stripe(rank, world)
    bn0 = hb.blue_noise.clone()

    tstream = torch.cuda.current_stream(dev)

    def one_step():
        if world == 1:
            # the next `spp` iterations of the progressive render (iteration indices, blue-noise table and running means
            # carry on from the previous step, as consecutive frames of the reference do): no host round trip between steps
            hb.render(spp)
            return
        hb.sync()                                       # the previous step's kernels are done with the blue-noise table
        hb.blue_noise.copy_(bn0)
        tstream.synchronize()                           # torch's stream -> visible to the ctx stream
        if bn_pre:
            hb.ctx.blue_noise_advance(hb.blue_noise, bn_pre, sd.width * sd.height)
        hb.render(spp, iter_stride=stride, iteration=first_it)
        if world > 1:
            hb.sync()                                   # ctx stream -> visible to torch's stream
            pkg.dist.combine_means(hb.accum, spp)
            tstream.synchronize()                       # the next step's kernels overwrite accum

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)
    hb.kp.iteration = first_it
    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
        if world == 1:
            pass
    fence()
    elapsed = time.perf_counter() - t0
    # per-kernel HIP-event times of the LAST step (events live on the ctx stream the kernels run on)
    st = hb.ctx.stats()
    trace_ms, resolve_ms, raygen_ms, tail_ms = st.trace_ms, st.resolve_ms, st.raygen_ms, st.tail_ms
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    samples_per_step_rank = W * H * spp
    total_samples = samples_per_step_rank * world * args.steps
    value = total_samples / elapsed / 1e6

    out = None
    if rank == 0:
        # ---- algorithmic bytes: counted on an untimed 2-iteration pass
        hb.ctx.set_counting(True)
        hb.blue_noise.copy_(bn0)
        hb.render(2, iter_stride=stride, iteration=first_it)
        hb.sync()
        cs = hb.ctx.stats()
        hb.ctx.set_counting(False)
        n = float(cs.samples)
        nd, nc, ne = cs.density_lookups / n, cs.color_lookups / n, cs.emission_lookups / n
        b_trace = 32.0 * nd + 128.0 * nc + 32.0 * ne + 64.0     # + the 64-byte path record the trace kernel writes
        b_survey = 32.0 * nd + 128.0 * nc + 32.0 * ne + 88.0    # SURVEY 8d figure (framebuffer term belongs to resolve)
        kernel_name = "vpt::trace_vol_kernel" if sd.kp.integrator else "vpt::trace_kernel"
        trace_s = trace_ms * 1e-3
        achieved = b_trace * samples_per_step_rank / trace_s / 1e9 if trace_s > 0 else 0.0
        roofline = {
            "bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
            "algorithmic_bytes_per_sample": round(b_trace, 2), "survey_8d_bytes_per_sample": round(b_survey, 2),
            "density_lookups_per_sample": round(nd, 4), "tracking_steps_per_sample": round(cs.tracking_steps / n, 4),
            "skip_steps_per_sample": round(cs.skip_steps / n, 4),
            "raygen_ms_per_step": round(raygen_ms, 3), "trace_ms_per_step": round(trace_ms, 3),
            "tail_resolve_ms_per_step": round(tail_ms + resolve_ms, 3),
            "rays_traced_fraction": round(cs.queued_rays / float(W * H * min(2, spp)), 4),
            "note": "achieved = algorithmic bytes (SURVEY 8d) x samples per launch / HIP-event time of the launch; traffic = HBM bytes per launch from "
                    "the committed rocprofv3 PMC passes (profiles/), null when this configuration has not been profiled",
        }
        # measured HBM traffic of the dominant kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
        # very command, corrected per MI355X_MICROARCH.md; tools/profile_bench.sh writes the file)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f).get(cfg)
            if tj and tj.get("width") == W and tj.get("height") == H:
                per_sample = tj["bytes_per_launch"] / float(tj["samples_per_launch"])
                # iterations per tracer launch: the chunk rule of vpt_render_batch (<= 64 iterations, <= 16 GiB of records)
                ipl = max(1, min(64, spp, (16 << 30) // (W * H * 64)))
                launches = -(-spp // ipl)
                roofline["traffic"] = round(per_sample * samples_per_step_rank / launches / 1e9, 4)
                roofline["launches_per_step"] = launches
                vk = (tj.get("valu") or {}).get(kernel_name)
                if vk:      # what bounds this kernel in practice: vector-instruction issue, at this many active lanes
                    roofline["valu_issue_busy"] = vk["valu_issue_busy"]
                    roofline["active_lanes_per_valu_instruction"] = vk["active_lanes_per_valu_instruction"]
                roofline["traffic_unit"] = "GB per launch (FETCH_SIZE x2 + WRITE_SIZE, %s)" % tj.get("source", "profiles/")
        cpu = None
        host_grids = all(isinstance(v[1], np.ndarray) for v in sd.volumes)
        if not args.no_cpu_baseline and world == 1 and host_grids:
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
            what = ("reference render_kernel.cu built for the host (oracle/_ref, %d threads over pixels)" % cores) if use_ref \
                else "oracle (OpenMP over rows)"
            cpu = {"value": round(W * H * args.cpu_iters / dtc / 1e6, 4), "unit": "Msamples/s", "cores": cores,
                   "kind": "reference" if use_ref else "port",
                   "sample": "%d of %d iterations of the same %dx%d frame, %s, %.1f s" % (args.cpu_iters, spp, W, H, what, dtc)}
        out = {
            "metric": "Msamples/s (W*H*spp/s)", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": data,
            "config": {"workload": workload, "width": W, "height": H, "spp_per_gpu": spp, "parallelism": "iteration-striped x%d + 1 all-reduce" % world,
                       "arithmetic": "strict (no FMA contraction, fixed-sequence log/sin/cos)"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
