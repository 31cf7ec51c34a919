#!/usr/bin/env python3
"""How coherent are the density look-ups a wave issues together?  (the evidence for / against
staging brick tiles in LDS.)   python tools/coherence_probe.py --config c4 [--grid-scale 1.0] [--spp 2]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c4"); ap.add_argument("--spp", type=int, default=2); ap.add_argument("--grid-scale", type=float, default=1.0)
a = ap.parse_args()
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
S = pkg.scene
if a.config == "c3":
    sd = S.fireball_scene(1920, 1080, n=256)
elif a.config == "c4":
    shape = tuple(int(round(x * a.grid_scale)) for x in (1216, 704, 1024))
    sd = S.cloud_scene(1920, 1080, env=(2048, 1024), integrator=1, device_grid=S.cloud_grid_torch(shape, device="cuda"))
else:
    sd = S.dragon_scene(1920, 1080, a.config)
if getattr(sd.kp, "sky_mult", 0) > 0 or a.config in ("c2", "c4"):
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
hb = S.HipBinding(sd, device=0)
hb.ctx.set_counting(True)
hb.render(a.spp, iteration=0); hb.sync()
lib = pkg.load_library()
coh = (C.c_ulonglong * 8)(); lib.vpt_test_get_coherence(hb.ctx.h, coh); c = list(coh)
sch = (C.c_ulonglong * 12)(); lib.vpt_test_get_schedule(hb.ctx.h, sch); o = list(sch)
st = hb.ctx.stats()
ev = max(1, c[0])
print("%s: %d sampled gather events; per event: %.1f lanes fetch, %.1f distinct 8^3 bricks (%.2f lanes per brick), %.1f distinct 4^3 bricks, "
      "%.2f lines per lane (own taps), %.1f distinct 128-B lines per event (%.2f per lane)"
      % (a.config, c[0], c[1] / ev, c[2] / ev, c[1] / max(1, c[2]), c[3] / ev, c[4] / max(1, c[1]), c[5] / ev, c[5] / max(1, c[1])))
print("traced rays that crossed empty nodes only (no draw, no look-up): %d of %d queued (%.1f %%)" % (c[6], st.queued_rays, 100.0 * c[6] / max(1, st.queued_rays)))
if o[0]:
    print("schedule: per pass walking %.1f, parked-in-T %.1f, idle %.1f lanes; tracking-step lanes %.1f/pass; density look-ups per sample %.2f"
          % (o[1] / o[0], o[2] / o[0], o[3] / o[0], o[7] / o[0], st.density_lookups / max(1, st.samples)))
