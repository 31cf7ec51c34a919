#!/usr/bin/env python3
"""Lane-bound tracer vs the round-3 pool tracer of a study library (VPT_LIB_PATH=.../libvpt_hip_pool.so, built by
`python volumetric-path-tracer_amd/build.py --variant pool --with-pool`): four scenes, every buffer and count bit-identical.
Run by tests/test_gpu_edge.py::test_pool_tracer_is_bit_identical_to_lane_tracer where that library exists."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

pkg = ge.load_package()


def make(scene):
    if scene == "dragon":
        return pkg.scene.dragon_scene(160, 90, "sun")
    if scene == "fireball":
        return pkg.scene.fireball_scene(96, 64, n=37)                 # emission march
    if scene == "instanced":
        return pkg.scene.instanced_scene(96, 64, n=18, grid=3, aperture=0.3)    # colour grids, open lens (primary ray re-read)
    sd = pkg.scene.dragon_scene(128, 72, "c1")                        # point light + the reference sphere in view
    sd.kp.ray_depth = 3
    sd.kp.volume_depth = 2
    return sd


def render(sd, counting):
    h = pkg.scene.HipBinding(sd, device=0)
    h.ctx.set_counting(counting)
    h.render(5)
    h.sync()
    return h, h.ctx.stats()


for scene in ("dragon", "fireball", "instanced", "sphere_lights"):
    sd = make(scene)
    os.environ.pop("VPT_TRACER", None)
    a, sa = render(sd, True)
    os.environ["VPT_TRACER"] = "pool"                                   # read when a context is created
    b, sb = render(sd, True)
    assert float(a.accum.abs().max()) > 0
    for buf in ("accum", "depth", "raw", "display"):
        np.testing.assert_array_equal(getattr(a, buf).cpu().numpy(), getattr(b, buf).cpu().numpy(), err_msg=scene + " " + buf)
    for k in ("samples", "density_lookups", "color_lookups", "emission_lookups", "tracking_steps", "skip_steps", "queued_rays"):
        assert getattr(sa, k) == getattr(sb, k), (scene, k)
    c, _ = render(sd, False)                                          # the non-counting instantiation too (13 rays per lane index instead of 12)
    np.testing.assert_array_equal(a.accum.cpu().numpy(), c.accum.cpu().numpy(), err_msg=scene)
print("4 scenes bit-identical")
