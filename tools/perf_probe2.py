#!/usr/bin/env python3
"""perf probe over bench scenes: python tools/perf_probe2.py --config c3 --spp 16 --set VPT_TRANS_MIN=8,32"""
import argparse, itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3"); ap.add_argument("--spp", type=int, default=8); ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--count-spp", type=int, default=4, help="iterations of the COUNTED render the schedule figures come from (64 = one full record chunk)")
ap.add_argument("--grid-scale", type=float, default=0.5, help="c4: linear scale of the 1024x704x1216 grid")
a = ap.parse_args()
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
S = pkg.scene
if a.config == "c3": sd = S.fireball_scene(1920, 1080, n=256)
elif a.config == "c5": sd = S.instanced_scene(3840, 2160, n=128, grid=10, aperture=2.0)
elif a.config == "c4":
    sd = S.cloud_scene(1920, 1080, env=(2048, 1024), integrator=1, device_grid=S.cloud_grid_torch(tuple(int(round(x * a.grid_scale)) for x in (1216, 704, 1024)), device="cuda"))
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
else:
    sd = S.dragon_scene(1920, 1080, a.config)
    if a.config == "c2": pkg.atmosphere.attach_default_atmosphere(sd, device=0)
keys = [s.split("=")[0] for s in a.set]; vals = [s.split("=")[1].split(",") for s in a.set]
for combo in itertools.product(*vals) if vals else [()]:
    for k, v in zip(keys, combo): os.environ[k] = v
    hb = S.HipBinding(sd, device=0)
    best = None
    for _ in range(a.reps):
        hb.render(a.spp, iteration=0); hb.sync(); st = hb.ctx.stats()
        tot = st.raygen_ms + st.trace_ms + st.tail_ms
        if best is None or tot < best[0]: best = (tot, st.raygen_ms, st.trace_ms, st.tail_ms)
    n = sd.width * sd.height * a.spp
    print(a.config, " ".join("%s=%s" % kv for kv in zip(keys, combo)), "raygen %.3f trace %.3f tail %.3f ms -> %.1f Msamples/s" % (best[1], best[2], best[3], n / best[0] / 1e3), flush=True)
    hb.ctx.close()

import ctypes as C
hb = S.HipBinding(sd, device=0)
hb.render(a.spp, iteration=0); hb.sync()
outp = (C.c_ulonglong * 12)()
pkg.load_library().vpt_test_get_schedule(hb.ctx.h, outp)
op = list(outp)
if sum(op[8:12]):
    tot = float(sum(op[8:12]))
    print("section cycles (non-counting kernel): refill %.1f%%, philox top-up %.1f%%, walk step %.1f%%, transitions %.1f%%" % tuple(100.0 * x / tot for x in op[8:12]))
    print("  of the walk step, the empty-node skip loop: %.1f%% of all cycles" % (100.0 * op[5] / tot))
    print("  transitions split: entry+FIRST_DONE %.1f%%, TRACK_DONE..EMIT %.1f%%, OUTER_SECOND/TOP %.1f%%, FINISH %.1f%%, Tr prologue %.1f%% (of all cycles)" % tuple(100.0 * x / tot for x in op[0:5]))
    co = (C.c_ulonglong * 8)()
    pkg.load_library().vpt_test_get_coherence(hb.ctx.h, co)
    co = list(co)
    if sum(co[0:4]):
        print("  refill split (of all cycles): idle test %.1f%%, claim (atomic + queue entries, waited for) %.1f%%, wait for the records %.1f%%, unpack %.1f%% | refills %d with %.1f lanes each, %d claims; cycles per refill: wait %.0f unpack %.0f, per claim %.0f"
              % (100.0 * co[0] / tot, 100.0 * co[1] / tot, 100.0 * co[2] / tot, 100.0 * co[3] / tot, co[4], co[5] / max(1, co[4]), co[6], co[2] / max(1, co[4]), co[3] / max(1, co[4]), co[1] / max(1, co[6])))
        print("  (idle test: with what the last pass left in flight; of a claim, its atomic alone: %.0f cycles)" % (co[7] / max(1, co[6])))
hb.ctx.set_counting(True)
hb.render(a.count_spp, iteration=0); hb.sync()
print("(schedule figures: a counted render of %d iterations)" % a.count_spp)
out = (C.c_ulonglong * 12)()
pkg.load_library().vpt_test_get_schedule(hb.ctx.h, out)
o = list(out); st = hb.ctx.stats()
if o[0]:
    print("schedule: passes %d | per pass: walking %.1f, parked-in-T %.1f, idle %.1f lanes | trans passes %.3f/pass with %.1f lanes each (%.2f inner passes) | tracking-step lanes %.1f/pass"
          % (o[0], o[1] / o[0], o[2] / o[0], o[3] / o[0], o[4] / o[0], o[6] / max(1, o[5]), o[5] / max(1, o[4]), o[7] / o[0]))
    print("per traced ray: passes*64/rays = %.1f lane-passes, steps %.2f, skips %.2f" % (o[0] * 64 / max(1, st.queued_rays), st.tracking_steps / max(1, st.queued_rays), st.skip_steps / max(1, st.queued_rays)))
rs = (C.c_ulonglong * 4)()
pkg.load_library().vpt_test_get_retry_stats(hb.ctx.h, rs)
rs = list(rs)
if sum(rs):
    tot = float(rs[0] + rs[1] + rs[3])
    print("vol_integrator's delta-tracking walks, lane-passes through the step proper: %.1f%% reach a look-up, %.1f%% end in retry spins only, %.1f%% end the walk (t >= distance, no retry left); retry draws per lane-pass %.2f (%.2f per retry-only pass)"
          % (100.0 * rs[0] / tot, 100.0 * rs[1] / tot, 100.0 * rs[3] / tot, rs[2] / tot, rs[2] / max(1.0, float(rs[1]))))
