#!/usr/bin/env python3
"""GPU box: do back-to-back SHORT batches of independent frames overlap when they alternate between two contexts (two streams, two sets of buffers)?
A k-iteration launch of the persistent tracer pays ~0.15-0.25 ms of start-up and drain; the next frame's kernels on another stream can run
under that drain.  Per k: ONE context rendering 2 n frames back to back (no host sync in between) against TWO contexts rendering n frames each, alternately.

    python tools/short_overlap_probe.py [--spps 1,4,8,16,64] [--frames 24]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--spps", default="4,8,16,64")
ap.add_argument("--frames", type=int, default=24)
args = ap.parse_args()
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
sd = pkg.scene.dragon_scene(1920, 1080, "c2")
pkg.atmosphere.attach_default_atmosphere(sd, device=0)
a = pkg.scene.HipBinding(sd, device=0)
b = pkg.scene.HipBinding(sd, device=0)
c = pkg.scene.HipBinding(sd, device=0)
for spp in [int(x) for x in args.spps.split(",")]:
    n = args.frames
    for hb in (a, b, c):
        hb.render(spp, iteration=0); hb.render(spp, iteration=0); hb.sync()
    best = {"one": 1e9, "two": 1e9, "three": 1e9}
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            a.render(spp, iteration=0)                    # every frame a fresh render (iteration 0: the running mean restarts), as a strong-scaling step is
        torch.cuda.synchronize()
        best["one"] = min(best["one"], (time.perf_counter() - t) / n)
        t = time.perf_counter()
        for i in range(n):
            (a if i % 2 == 0 else b).render(spp, iteration=0)
        torch.cuda.synchronize()
        best["two"] = min(best["two"], (time.perf_counter() - t) / n)
        t = time.perf_counter()
        for i in range(n):
            (a, b, c)[i % 3].render(spp, iteration=0)
        torch.cuda.synchronize()
        best["three"] = min(best["three"], (time.perf_counter() - t) / n)
    px = 1920 * 1080 * spp
    print("spp %2d: one context %.4f ms/frame (%.0f Ms/s) | two alternating %.4f ms/frame (%.0f Ms/s, x%.3f) | three %.4f ms/frame (%.0f Ms/s, x%.3f)" %
          (spp, best["one"] * 1e3, px / best["one"] / 1e6, best["two"] * 1e3, px / best["two"] / 1e6, best["one"] / best["two"],
           best["three"] * 1e3, px / best["three"] / 1e6, best["one"] / best["three"]), flush=True)
