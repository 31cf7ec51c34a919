#!/usr/bin/env python3
"""vol_integrator with the PROCEDURAL sky as its light (environment_type 0: estimate_sky evaluates the Bruneton sky inside the tracer -- the SKYLUT
instantiations of trace_vol_kernel, which no BASELINE config runs): step time of the dragon at 1080p, for A/B of library variants (VPT_LIB_PATH).
    python tools/vol_sky_probe.py [--spp 8] [--steps 4]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
pkg = ge.load_package()
sd = pkg.scene.dragon_scene(a.width, a.height, "c2")
sd.kp.integrator = 1
sd.kp.ray_depth = 6
sd.kp.density_mult = 2.0
sd.env_cdf = pkg.host.env_cdf_build(sd.kp)
pkg.atmosphere.attach_default_atmosphere(sd, device=0)
hb = pkg.scene.HipBinding(sd, device=0)
hb.render(a.spp); hb.sync()
t0 = time.perf_counter()
for _ in range(a.steps):
    hb.render(a.spp)
hb.sync()
dt = (time.perf_counter() - t0) / a.steps
st = hb.ctx.stats()
print("%s: vol_integrator + procedural sky, %dx%d x %d spp: step %.3f ms (last step: raygen %.3f trace %.3f tail %.3f) -> %.1f Msamples/s" % (
    os.path.basename(os.environ.get("VPT_LIB_PATH", "libvpt_hip.so")), a.width, a.height, a.spp, dt * 1e3, st.raygen_ms, st.trace_ms, st.tail_ms,
    a.width * a.height * a.spp / dt / 1e6))
