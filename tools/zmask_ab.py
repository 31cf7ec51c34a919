#!/usr/bin/env python3
"""The zero-footprint mask of a study library (VPT_LIB_PATH=.../libvpt_hip_zmask.so, built by
`python volumetric-path-tracer_amd/build.py --variant zmask -DVPT_ZERO_MASK`) against the same library without it: four scenes x two block
edges, every buffer and every count bit-identical, the mask really answers look-ups, the image is the oracle's.
Run by tests/test_gpu_edge.py::test_zero_footprint_mask_is_bit_identical where that library exists.  (Round 6: exact, and slower than the two
loads it saves on every config -- profiles/r06_zero_mask.txt -- so the product library does not carry it.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import oracle_binding

pkg = ge.load_package()


def make(scene):
    if scene == "dragon":
        return pkg.scene.dragon_scene(96, 64, "sun")
    if scene == "fireball":
        return pkg.scene.fireball_scene(96, 64, n=37)
    if scene == "cloud_vol":
        sd = pkg.scene.cloud_scene(96, 64, shape=(45, 31, 38), env=(64, 32))        # vol_integrator: the split-phase look-up
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
        return sd
    return pkg.scene.instanced_scene(96, 64, n=18, grid=3, aperture=0.3)


def render(sd, counting):
    h = pkg.scene.HipBinding(sd, device=0)
    h.ctx.set_counting(counting)
    h.render(3)
    h.sync()
    return h, h.ctx.stats()


def rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(1e-30, np.sqrt((b ** 2).sum())))


cases = 0
for scene in ("dragon", "fireball", "instanced", "cloud_vol"):
    sd = make(scene)
    for k in ("VPT_ZERO_MASK", "VPT_ZERO_MASK_MIN_BYTES", "VPT_ZERO_MASK_SHIFT"):
        os.environ.pop(k, None)
    a, sa = render(sd, True)
    assert float(a.accum.abs().max()) > 0 and sa.density_zero_skips == 0
    ob = oracle_binding.OracleBinding(sd)
    ob.render(3)
    for shift in ("2", "3"):
        os.environ.update(VPT_ZERO_MASK="1", VPT_ZERO_MASK_MIN_BYTES="0", VPT_ZERO_MASK_SHIFT=shift)     # read when a context is created
        b, sb = render(sd, True)
        for buf in ("accum", "depth", "raw", "display", "cost"):
            np.testing.assert_array_equal(getattr(a, buf).cpu().numpy(), getattr(b, buf).cpu().numpy(), err_msg="%s %s shift %s" % (scene, buf, shift))
        for k in ("samples", "density_lookups", "color_lookups", "emission_lookups", "tracking_steps", "skip_steps", "queued_rays",
                  "density_fetches", "color_fetches", "emission_fetches"):
            assert getattr(sa, k) == getattr(sb, k), (scene, shift, k)
        assert 0 < sb.density_zero_skips < sb.density_fetches, (scene, shift, sb.density_zero_skips, sb.density_fetches)
        c, _ = render(sd, False)                                     # the timed (non-counting) instantiation takes the same path
        np.testing.assert_array_equal(a.accum.cpu().numpy(), c.accum.cpu().numpy(), err_msg=scene)
        np.testing.assert_array_equal(a.depth.cpu().numpy(), c.depth.cpu().numpy(), err_msg=scene)
        np.testing.assert_array_equal(b.depth.cpu().numpy(), ob.depth)
        assert rel_l2(b.accum.cpu().numpy(), ob.accum) <= (1e-3 if scene == "cloud_vol" else 2e-6)
        print("%-10s block edge %d: zero-mask answered %d of %d density fetches" % (scene, 1 << int(shift), sb.density_zero_skips, sb.density_fetches))
        cases += 1
print("%d cases bit-identical" % cases)
