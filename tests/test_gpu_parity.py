"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Tolerance of BASELINE.json's north star: relative L2 <= 1e-3 per image at
fixed RNG seeds.  Because both sides evaluate the decision path in the same strict arithmetic,
the tests also assert the much tighter figure actually achieved."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_L2_TOL = 1e-3        # north-star tolerance (BASELINE.json)
REL_L2_TIGHT = 2e-6      # what strict arithmetic delivers (value-only libm differences in exp())


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(1e-30, np.sqrt((b ** 2).sum())))


@pytest.fixture(scope="module")
def ob_mod():
    import oracle_binding
    return oracle_binding


def _pair(pkg, ob_mod, w, h, config, tweak=None):
    sd = pkg.scene.dragon_scene(w, h, config)
    if tweak:
        tweak(sd)
    hb = pkg.scene.HipBinding(sd, device=0)
    ob = ob_mod.OracleBinding(sd)
    return sd, hb, ob


def test_device_math_and_rng_match_oracle_bitwise(pkg, orc):
    """The fixed-sequence log/sin/cos, fp32 divide/sqrt and the rocRAND Philox stream produce on
    gfx950 the very bits the oracle computes on the host."""
    ctx = pkg.Context(0)
    lib = pkg.load_library()
    rng = np.random.default_rng(5)
    xs = np.concatenate([1.0 - rng.uniform(0, 1, 100000), 2.0 ** -rng.uniform(0, 40, 5000), [0.0, 1.0]]).astype(np.float32)
    out = np.zeros_like(xs)
    assert lib.vpt_test_device_math(ctx.h, 0, xs.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), len(xs)) == 0
    ref = np.array([orc.orc_det_logf(float(x)) for x in xs], np.float32)
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))
    ang = rng.uniform(0, 2 * np.pi, 100000).astype(np.float32)
    for op, f in ((1, orc.orc_det_sinf), (2, orc.orc_det_cosf)):
        o = np.zeros_like(ang)
        assert lib.vpt_test_device_math(ctx.h, op, ang.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), len(ang)) == 0
        r = np.array([f(float(a)) for a in ang], np.float32)
        np.testing.assert_array_equal(o.view(np.uint32), r.view(np.uint32))
    # correctly rounded divide / sqrt (HIP default) == IEEE host results
    v = rng.uniform(1e-6, 1e6, 100000).astype(np.float32)
    o = np.zeros_like(v)
    assert lib.vpt_test_device_math(ctx.h, 4, v.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), len(v)) == 0
    np.testing.assert_array_equal(o, np.float32(1.0) / v)
    assert lib.vpt_test_device_math(ctx.h, 5, v.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), len(v)) == 0
    np.testing.assert_array_equal(o, np.sqrt(v))
    # Philox stream incl. offsets that are not multiples of 4
    for seed, off in ((0, 0), (12345, 4096 * 7), (99, 4096 * 3 + 5)):
        a = np.zeros(41, np.float32); b = np.zeros(41, np.float32)
        assert lib.vpt_test_device_uniform_stream(ctx.h, seed, off, 41, a.ctypes.data_as(C.c_void_p)) == 0
        orc.orc_curand_uniform_stream(seed, off, 41, b.ctypes.data_as(C.c_void_p))
        np.testing.assert_array_equal(a, b)
        # the product's own Philox (csrc/vpt_rng.h) through its refill-point protocol
        c = np.zeros(41, np.float32)
        assert lib.vpt_test_device_product_stream(ctx.h, seed, off, 41, c.ctypes.data_as(C.c_void_p)) == 0
        np.testing.assert_array_equal(c, b)
    ctx.close()


def test_octree_builder_matches_oracle(pkg, orc, ob_mod):
    sd, hb, ob = _pair(pkg, ob_mod, 16, 16, "c1")
    lo, hi, mx, mn = hb.ctx.root()
    info = ob_mod.OctreeInfo()
    assert orc.orc_octree_info_get(ob.volumes, 1, C.byref(info)) == 0
    assert lo == info.root_pmin.tuple() and hi == info.root_pmax.tuple()
    assert (mx, mn) == (info.max_extinction, info.min_extinction)
    assert hb.ctx.octree_stats() == list(info.nonempty)


@pytest.mark.parametrize("config,spp", [("c1", 4), ("sun", 3)])
def test_image_parity_dragon(pkg, ob_mod, config, spp):
    """BASELINE config 1 (point light, no atmosphere) and the sun-only variant of config 2."""
    sd, hb, ob = _pair(pkg, ob_mod, 160, 120, config)
    hb.ctx.set_counting(True)
    hb.render(spp)
    hb.sync()
    ob.render(spp)
    got = hb.accum.cpu().numpy()
    assert np.isfinite(got).all()
    assert ob.accum.max() > 0
    e = rel_l2(got, ob.accum)
    assert e <= REL_L2_TOL, e
    assert e <= REL_L2_TIGHT, e
    # depth buffer, alpha (raw.w) and the 8-bit display
    np.testing.assert_allclose(hb.depth.cpu().numpy(), ob.depth, rtol=1e-6, atol=1e-6)
    raw = hb.raw.cpu().numpy()
    np.testing.assert_allclose(raw[:, 3], ob.raw[:, 3], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(raw[:, :3], ob.raw[:, :3], rtol=1e-4, atol=1e-5)
    disp = hb.display.cpu().numpy().view(np.uint32)
    diff = np.abs(((disp[:, None] >> np.array([16, 8, 0])) & 255).astype(int) - ((ob.display[:, None] >> np.array([16, 8, 0])) & 255).astype(int))
    assert diff.max() <= 1 and (disp >> 24 == 255).all()
    # blue-noise buffer advanced exactly like `spp` launches would
    np.testing.assert_array_equal(hb.blue_noise.cpu().numpy(), ob.blue_noise)
    # the random walks are identical: exact look-up / step / skip counts
    st = hb.ctx.stats()
    assert st.samples == ob.stats.samples == 160 * 120 * spp
    assert st.density_lookups == ob.stats.density_lookups
    assert st.tracking_steps == ob.stats.tracking_steps
    assert st.skip_steps == ob.stats.skip_steps


@pytest.mark.parametrize("seed", [1, 2])
def test_random_setups_vs_oracle(pkg, ob_mod, seed):
    """The parity tests above and in test_gpu_edge.py use hand-built scenes; this one draws them: SEEDED RANDOM camera (around and sometimes inside the dragon's box), reference
    sphere (from overlapping the box's edge to far away, small and large; above the ground: a view point inside the planet is outside the sky model's domain), sun direction, ray_depth 1-4, volume_depth 1-3, density_mult x 0.25 / 1 / 3, phase g, the volume
    rotated about y in a third of the cases, point lights in a quarter, the procedural sky in a quarter, odd image extents.  Timed HIP render (every shortcut on) against
    the oracle: depth BIT-identical (a function of the walk decisions alone), accum within north_star's 1e-3 (2e-6 without the sky's value-only tables); counting render:
    look-up / step / skip counts equal to the oracle's."""
    from random_setups import dragon_setup
    rs = np.random.RandomState(500 + seed)
    for case in range(8):
        sd, w, h, sky, desc = dragon_setup(pkg, rs, case)
        spp = 3
        ob = ob_mod.OracleBinding(sd)
        ob.render(spp, nthreads=os.cpu_count() or 1)
        for counting in (False, True):
            hb = pkg.scene.HipBinding(sd, device=0)
            hb.ctx.set_counting(counting)
            hb.render(spp)
            hb.sync()
            got, dgot = hb.accum.cpu().numpy(), hb.depth.cpu().numpy()
            st = hb.ctx.stats()
            hb.ctx.close()
            assert np.isfinite(got).all()
            np.testing.assert_array_equal(dgot, ob.depth, err_msg="seed %d case %d (counting %s): depth" % (seed, case, counting))
            e = rel_l2(got, ob.accum)
            assert e <= (REL_L2_TOL if sky else REL_L2_TIGHT), (seed, case, counting, e)
            if counting:
                assert st.samples == ob.stats.samples == w * h * spp
                assert (st.density_lookups, st.tracking_steps, st.skip_steps) == (ob.stats.density_lookups, ob.stats.tracking_steps, ob.stats.skip_steps), (seed, case)
        print("seed %d case %d: %d x %d, rel L2 %.2e, %d steps, %d skips" % (seed, case, w, h, e, ob.stats.tracking_steps, ob.stats.skip_steps))


@pytest.mark.parametrize("seed", [1, 2])
def test_random_kernel_params_vs_oracle(pkg, ob_mod, seed):
    """The Kernel_params fields the set-ups above leave at their defaults, drawn at random on the dragon (sun or point lights + sun, both integrators): albedo / extinction
    per channel, tr_depth, energy_inject, the sun's colour and multiplier, exposure, a first iteration other than 0 and an iteration stride (striping), max_interactions inside
    the batch (the iterations past it are not rendered: WHITE), image extents that are not multiples of the 8 x 8 / 64 x 64 tiles.  Depth bit-identical, counts equal,
    accum within 2e-6 (1e-3 for the vol_integrator cases: their tail is the procedural sky); the blue-noise state advances as the oracle's."""
    from vpt_amd.abi import Float3
    from random_setups import random_view
    rs = np.random.RandomState(900 + seed)
    for case in range(8):
        w, h = int(rs.choice([101, 127, 64, 90])), int(rs.choice([59, 71, 36]))
        sd = pkg.scene.dragon_scene(w, h, "c1" if case % 2 else "sun")
        if case % 2:
            sd.kp.sun_mult = float(rs.uniform(0.0, 2.0))
        vol = case % 4 == 2
        if vol:
            sd.kp.integrator = 1                                    # (its tail is always the procedural sky, render_kernel.cu:1752: the tables must be bound)
            pkg.atmosphere.attach_default_atmosphere(sd, device=0)
        desc = random_view(pkg, rs, sd, w, h, above_ground=vol)
        a = rs.uniform(0.2, 1.0, 3); e = rs.uniform(0.3, 1.0, 3)
        sd.kp.albedo = Float3(float(a[0]), float(a[1]), float(a[2]))
        sd.kp.extinction = Float3(float(e[0]), float(e[1]), float(e[2]))
        sd.kp.tr_depth = float(rs.choice([0.5, 1.0, 2.0]))
        sd.kp.energy_inject = float(rs.choice([0.0, 0.5, 3.0]))
        c = rs.uniform(0.2, 1.0, 3)
        sd.kp.sun_color = Float3(float(c[0]), float(c[1]), float(c[2]))
        sd.kp.exposure_scale = float(rs.uniform(0.3, 3.0))
        spp, it0, stride = int(rs.randint(2, 5)), int(rs.randint(0, 40)), int(rs.choice([1, 1, 2, 5]))
        if case == 5:
            sd.kp.max_interactions = it0 + stride                   # the batch's first iteration renders, the later ones are past the limit
        ob = ob_mod.OracleBinding(sd)
        ob.render(spp, iter_stride=stride, iteration=it0, nthreads=os.cpu_count() or 1)
        for counting in (False, True):
            hb = pkg.scene.HipBinding(sd, device=0)
            hb.ctx.set_counting(counting)
            hb.render(spp, iteration=it0, iter_stride=stride)
            hb.sync()
            got, dgot, bn = hb.accum.cpu().numpy(), hb.depth.cpu().numpy(), hb.blue_noise.cpu().numpy()
            st = hb.ctx.stats()
            hb.ctx.close()
            assert np.isfinite(got).all()
            np.testing.assert_array_equal(dgot, ob.depth, err_msg="seed %d case %d (counting %s): depth | %s" % (seed, case, counting, desc))
            np.testing.assert_array_equal(bn, ob.blue_noise)
            e2 = rel_l2(got, ob.accum)
            assert e2 <= (REL_L2_TOL if vol else REL_L2_TIGHT), (seed, case, counting, e2, desc)
            if counting:
                assert st.samples == ob.stats.samples
                assert (st.density_lookups, st.tracking_steps, st.skip_steps) == (ob.stats.density_lookups, ob.stats.tracking_steps, ob.stats.skip_steps), (seed, case, desc)
        print("seed %d case %d: integrator %d, %d iterations from %d stride %d, rel L2 %.2e, %d steps | %s" % (seed, case, sd.kp.integrator, spp, it0, stride, e2, ob.stats.tracking_steps, desc))


@pytest.mark.parametrize("kind", ["fireball", "fireball sky", "instances", "instances open lens", "cloud vol_integrator"])
def test_random_views_of_the_other_scenes_vs_oracle(pkg, ob_mod, kind):
    """test_random_setups_vs_oracle for the other tracer instantiations: the emission march (fireball; sun only and with the procedural sky), instanced coloured volumes (the
    generic walk with per-leaf instance lists and the colour look-ups; closed lens and open lens: every sample has its own origin), the vol_integrator over a cloud with an HDRI
    (the second tracer kernel).  Four seeded random views each -- one from inside the volumes -- with random sun, loop depths, density, phase g and the sphere near the volumes
    half of the time.  Depth bit-identical, look-up / step / skip counts equal, accum within north_star's 1e-3 (2e-6 where no value-only sky code is involved)."""
    from random_setups import random_view
    rs = np.random.RandomState({"fireball": 21, "fireball sky": 22, "instances": 23, "instances open lens": 24, "cloud vol_integrator": 25}[kind])
    for view in range(4):
        w, h = int(rs.choice([112, 96, 83])), int(rs.choice([63, 54, 47]))
        sky = False
        aperture = 0.0
        if kind.startswith("fireball"):
            sky = kind.endswith("sky")
            sd = pkg.scene.fireball_scene(w, h, n=40, sky=sky)
        elif kind.startswith("instances"):
            aperture = float(rs.uniform(0.5, 3.0)) if kind.endswith("open lens") else 0.0
            sd = pkg.scene.instanced_scene(w, h, n=24, grid=3, aperture=aperture, sky=False)
        else:
            sky = True
            sd = pkg.scene.cloud_scene(w, h, shape=(40, 36, 48), env=(96, 48), integrator=1)
        if sky:
            pkg.atmosphere.attach_default_atmosphere(sd, device=0)
        desc = random_view(pkg, rs, sd, w, h, aperture=aperture, inside=view == 3, above_ground=sky)
        spp = 2
        ob = ob_mod.OracleBinding(sd)
        ob.render(spp, nthreads=os.cpu_count() or 1)
        for counting in (False, True):
            hb = pkg.scene.HipBinding(sd, device=0)
            hb.ctx.set_counting(counting)
            hb.render(spp)
            hb.sync()
            got, dgot = hb.accum.cpu().numpy(), hb.depth.cpu().numpy()
            st = hb.ctx.stats()
            hb.ctx.close()
            assert np.isfinite(got).all()
            np.testing.assert_array_equal(dgot, ob.depth, err_msg="%s view %d (counting %s): depth | %s" % (kind, view, counting, desc))
            e = rel_l2(got, ob.accum)
            assert e <= (REL_L2_TOL if sky else REL_L2_TIGHT), (kind, view, counting, e, desc)
            if counting:
                assert st.samples == ob.stats.samples == w * h * spp
                assert (st.density_lookups, st.color_lookups, st.emission_lookups, st.tracking_steps, st.skip_steps) == \
                       (ob.stats.density_lookups, ob.stats.color_lookups, ob.stats.emission_lookups, ob.stats.tracking_steps, ob.stats.skip_steps), (kind, view, desc)
        print("%s view %d: rel L2 %.2e, %d steps, %d look-ups | %s" % (kind, view, e, ob.stats.tracking_steps, ob.stats.density_lookups, desc))


def test_per_pixel_values_match_single_sample(pkg, ob_mod):
    """One iteration, pixel by pixel: accum after iteration 0 is the sample value itself."""
    sd, hb, ob = _pair(pkg, ob_mod, 96, 64, "c1", tweak=lambda s: setattr(s.kp, "sun_mult", 1.0))
    hb.render(1, iteration=5)
    hb.sync()
    got = hb.accum.cpu().numpy().reshape(64, 96, 3)
    rng = np.random.default_rng(7)
    bright = np.argwhere(got.sum(-1) > 0)
    picks = [tuple(p) for p in bright[rng.choice(len(bright), min(40, len(bright)), replace=False)]] + [(0, 0), (63, 95), (32, 48)]
    # iteration 5 on a fresh buffer is "local index 5" -> running mean step; compare sample values via the oracle's probe
    ob.kp.iteration = 5
    for (y, x) in picks:
        val = ob.sample_pixel(x, y, iteration=5)[:3]
        exp = (np.zeros(3, np.float32) + (val - 0) / np.float32(6)).astype(np.float32)
        np.testing.assert_allclose(got[y, x], exp, rtol=2e-6, atol=1e-9)


def test_batch_equals_repeated_single_launches(pkg, ob_mod):
    """vpt_render_batch(n) leaves accum/depth/blue-noise bit-identical to n calls of vpt_render."""
    sd = pkg.scene.dragon_scene(128, 72, "c1")
    a = pkg.scene.HipBinding(sd, device=0)
    b = pkg.scene.HipBinding(sd, device=0)
    a.render(5)
    for _ in range(5):
        b.render(1)
    a.sync(); b.sync()
    np.testing.assert_array_equal(a.accum.cpu().numpy(), b.accum.cpu().numpy())
    np.testing.assert_array_equal(a.depth.cpu().numpy(), b.depth.cpu().numpy())
    np.testing.assert_array_equal(a.blue_noise.cpu().numpy(), b.blue_noise.cpu().numpy())
    np.testing.assert_array_equal(a.display.cpu().numpy(), b.display.cpu().numpy())


def test_schedule_independence(pkg, monkeypatch):
    """A sample is a pure function of (pixel, iteration): changing the persistent grid size or
    the batch chunking must not change a single bit."""
    sd = pkg.scene.dragon_scene(200, 100, "sun")
    imgs = []
    for bpc, chunk in (("1", "1"), ("4", "3"), ("8", "64")):
        monkeypatch.setenv("VPT_BLOCKS_PER_CU", bpc)
        monkeypatch.setenv("VPT_BATCH_ITERS", chunk)
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.render(6)
        hb.sync()
        imgs.append(hb.accum.cpu().numpy())
        hb.ctx.close()
    np.testing.assert_array_equal(imgs[0], imgs[1])
    np.testing.assert_array_equal(imgs[0], imgs[2])


def test_iteration_striping_two_ranks(pkg, ob_mod):
    """Multi-GPU partition on one device: rank r renders iterations r, r+2, ...; the weighted
    combination equals the 1-rank image (fp32 summation order differs -> 1e-6)."""
    import torch
    sd = pkg.scene.dragon_scene(120, 80, "c1")
    full = pkg.scene.HipBinding(sd, device=0)
    full.render(8)
    full.sync()
    parts = []
    for r in range(2):
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.ctx.blue_noise_advance(hb.blue_noise, r, sd.width * sd.height)
        hb.render(4, iter_stride=2, iteration=r)
        hb.sync()
        parts.append(hb.accum.double() * 4)
    comb = ((parts[0] + parts[1]) / 8).float().cpu().numpy()
    assert rel_l2(comb, full.accum.cpu().numpy()) < 1e-6
    # and the oracle agrees with a striped rank
    ob = ob_mod.OracleBinding(sd)
    for _ in range(1):
        pass
    live = min(sd.width * sd.height, 65536)       # a launch advances only the entries its pixels own
    ob.blue_noise[:live] = np.mod(ob.blue_noise[:live] + np.float32((1 + np.sqrt(np.float32(5))) / 2), np.float32(1.0)).astype(np.float32)
    ob.render(4, iter_stride=2, iteration=1)
    assert rel_l2((parts[1] / 4).float().cpu().numpy(), ob.accum) < REL_L2_TIGHT


def test_full_hd_properties(pkg):
    """BASELINE's 1920x1080 frame, one iteration: finite, non-negative, deterministic across runs,
    every pixel whose primary ray misses the volume box is exactly black (no sky bound, no lights hit)."""
    sd = pkg.scene.dragon_scene(1920, 1080, "sun")
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(1)
    hb.sync()
    a = hb.accum.cpu().numpy().copy()
    assert np.isfinite(a).all() and (a >= 0).all() and a.max() > 0
    hb2 = pkg.scene.HipBinding(sd, device=0)
    hb2.render(1)
    hb2.sync()
    np.testing.assert_array_equal(a, hb2.accum.cpu().numpy())
    frac = float((a.sum(1) > 0).mean())
    assert 0.001 < frac < 0.5


def test_error_paths(pkg):
    """Error behaviour of the boundary: codes + messages instead of exit(1) (main.cpp:136-142)."""
    ctx = pkg.Context(0)
    sd = pkg.scene.dragon_scene(8, 8, "c1")
    kp = pkg.abi.KernelParams.from_buffer_copy(sd.kp)
    lights = pkg.abi.LightList(0, None)
    with pytest.raises(pkg.VptError, match="NOT_READY"):
        ctx.render(sd.camera, lights, sd.sphere, sd.atmosphere, kp)
    with pytest.raises(pkg.VptError, match="INVALID"):
        ctx.set_volumes([sd.volumes[0][0]])           # no density texture handle
    hb = pkg.scene.HipBinding(sd, device=0, ctx=ctx)
    bad = pkg.abi.KernelParams.from_buffer_copy(hb.kp)
    bad.sky_mult = 1.0                                 # sky requested, no LUTs bound
    with pytest.raises(pkg.VptError, match="NOT_READY"):
        ctx.render(sd.camera, hb.lights, sd.sphere, hb.atmosphere, bad)
    bad = pkg.abi.KernelParams.from_buffer_copy(hb.kp)
    bad.accum_buffer = None
    with pytest.raises(pkg.VptError, match="INVALID"):
        ctx.render(sd.camera, hb.lights, sd.sphere, hb.atmosphere, bad)

