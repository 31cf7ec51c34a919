#!/usr/bin/env python3
"""Does the GPU overlap two independent render pipelines (two contexts, two streams)?  Decides whether a chunk-level
software pipeline (raygen/tail of one chunk under the tracer of another) is worth building.

    python tools/overlap_probe.py [--spp 32]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=32)
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    pkg = ge.load_package()
    sd = pkg.scene.dragon_scene(1920, 1080, "c2")
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    for bpc in ("3", "2", "1"):
        os.environ["VPT_BLOCKS_PER_CU"] = bpc
        a = pkg.scene.HipBinding(sd, device=0)
        b = pkg.scene.HipBinding(sd, device=0)
        for hb in (a, b):
            hb.render(args.spp, iteration=0)
            hb.sync()
        best_seq = best_par = 1e9
        for _ in range(3):
            t = time.perf_counter()
            a.render(args.spp, iteration=0); a.sync()
            b.render(args.spp, iteration=0); b.sync()
            best_seq = min(best_seq, time.perf_counter() - t)
            t = time.perf_counter()
            a.render(args.spp, iteration=0)
            b.render(args.spp, iteration=0)
            a.sync(); b.sync()
            best_par = min(best_par, time.perf_counter() - t)
        n = 2 * 1920 * 1080 * args.spp
        print("blocks/CU %s: sequential %.3f ms (%.0f Ms/s)  concurrent %.3f ms (%.0f Ms/s)" %
              (bpc, best_seq * 1e3, n / best_seq / 1e6, best_par * 1e3, n / best_par / 1e6), flush=True)
        a.ctx.close(); b.ctx.close()


if __name__ == "__main__":
    main()
