#!/usr/bin/env python3
"""Quick perf probe (GPU box): trace/resolve kernel times for a grid of runtime knobs.

    python tools/perf_probe.py [--config c2] [--spp 16] --set VPT_REGEN_MIN=8,16,32 --set VPT_BLOCKS_PER_CU=2,3,4
"""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--set", action="append", default=[])
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    ge.build()
    pkg = ge.load_package()
    sd = pkg.scene.dragon_scene(args.width, args.height, args.config)
    if args.config == "c2":
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    keys = [s.split("=")[0] for s in args.set]
    vals = [s.split("=")[1].split(",") for s in args.set]
    for combo in itertools.product(*vals) if vals else [()]:
        for k, v in zip(keys, combo):
            os.environ[k] = v
        hb = pkg.scene.HipBinding(sd, device=0)
        best = None
        for _ in range(args.reps):
            hb.render(args.spp, iteration=0)
            hb.sync()
            st = hb.ctx.stats()
            tot = st.raygen_ms + st.trace_ms + st.tail_ms
            if best is None or tot < best[0]:
                best = (tot, st.raygen_ms, st.trace_ms, st.tail_ms, 0.0, st.queued_rays)
        n = args.width * args.height * args.spp
        print(" ".join("%s=%s" % kv for kv in zip(keys, combo)), "raygen %.3f trace %.3f tail %.3f resolve %.3f ms -> %.1f Msamples/s, queued %d of %d" %
              (best[1], best[2], best[3], best[4], n / best[0] / 1e3, best[5], n), flush=True)
        hb.ctx.close()


if __name__ == "__main__":
    main()
