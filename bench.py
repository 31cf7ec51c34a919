#!/usr/bin/env python3
"""bench.py -- Msamples/s of the hot path (BASELINE.json metric) on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|sun|c1] [--spp 64]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" = one render of the named workload: dragon.vdb, 1920x1080, `spp` iterations of
`volume_rt_kernel` per rank (trace + resolve kernels, inputs resident in HBM).  With N > 1
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
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=16)
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W, H, spp = args.width, args.height, args.spp
    sd = pkg.scene.dragon_scene(W, H, args.config)
    workload = "dragon.vdb %dx%dx%dspp, %s" % (W, H, spp, {
        "c2": "procedural sun+sky (BASELINE config 2)", "sun": "sun NEE only, sky_mult=0 (config 2 without the sky LUTs)",
        "c1": "one point light, no atmosphere (BASELINE config 1)"}[args.config])
    if args.config == "c2":
        pkg.atmosphere.attach_default_atmosphere(sd, device=local_rank)
    hb = pkg.scene.HipBinding(sd, device=local_rank)
    first_it, stride, bn_pre = pkg.dist.stripe(rank, world)
    bn0 = hb.blue_noise.clone()

    def one_step():
        hb.blue_noise.copy_(bn0)
        if bn_pre:
            hb.ctx.blue_noise_advance(hb.blue_noise, bn_pre)
        hb.render(spp, iter_stride=stride, iteration=first_it)
        if world > 1:
            hb.sync()                                   # ctx stream -> visible to torch's stream
            pkg.dist.combine_means(hb.accum, spp)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)
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
        trace_s = trace_ms * 1e-3
        achieved = b_trace * samples_per_step_rank / trace_s / 1e9 if trace_s > 0 else 0.0
        roofline = {
            "bound": "hbm", "kernel": "vpt::trace_kernel", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
            "algorithmic_bytes_per_sample": round(b_trace, 2), "survey_8d_bytes_per_sample": round(b_survey, 2),
            "density_lookups_per_sample": round(nd, 4), "tracking_steps_per_sample": round(cs.tracking_steps / n, 4),
            "skip_steps_per_sample": round(cs.skip_steps / n, 4),
            "raygen_ms_per_step": round(raygen_ms, 3), "trace_ms_per_step": round(trace_ms, 3),
            "tail_ms_per_step": round(tail_ms, 3), "resolve_ms_per_step": round(resolve_ms, 3),
            "rays_traced_fraction": round(cs.queued_rays / float(W * H * min(2, spp)), 4),
            "note": "dragon grid is 425 KB (L2-resident): the fraction is algorithmic bytes / HBM peak, not measured HBM traffic",
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding
            ob = oracle_binding.OracleBinding(sd)
            cores = os.cpu_count() or 1
            tc = time.perf_counter()
            ob.render(args.cpu_iters, nthreads=cores)
            dtc = time.perf_counter() - tc
            cpu = {"value": round(W * H * args.cpu_iters / dtc / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                   "sample": "%d of %d iterations of the same %dx%d frame, oracle (OpenMP over rows), %.1f s" % (args.cpu_iters, spp, W, H, dtc)}
        out = {
            "metric": "Msamples/s (W*H*spp/s)", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic camera/lights on the reference's dragon.vdb grid (committed fixture)",
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
