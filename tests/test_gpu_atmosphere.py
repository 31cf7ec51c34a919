"""Procedural sky (BASELINE config 2): sanity of the precomputed look-up tables and parity of the
full sun+sky render against the oracle fed with the SAME tables."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(1e-30, np.sqrt((b ** 2).sum())))


@pytest.fixture(scope="module")
def sky(pkg):
    sd = pkg.scene.dragon_scene(8, 8, "c2")
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    return sd.atmosphere, sd.atm_luts


def test_default_model_scalars(pkg):
    p = pkg.atmosphere.default_model()
    assert (p.bottom_radius, p.top_radius) == (6360000.0, 6420000.0)
    assert p.mie_phase_function_g == pytest.approx(0.8) and p.sun_angular_radius == pytest.approx(0.004675)
    assert p.mu_s_min == pytest.approx(-0.5, abs=1e-6) and p.use_luminance == 0 and p.exposure == 1.0
    np.testing.assert_allclose(p.rayleigh_scattering.tuple(), 1.24062e-6 * np.array([0.68, 0.55, 0.44]) ** -4.0, rtol=1e-6)
    np.testing.assert_allclose(p.mie_scattering.tuple(), [5.328e-3 / 1200 * 0.9] * 3, rtol=1e-6)
    assert p.mie_extinction.tuple() == p.mie_scattering.tuple()          # reference quirk D4
    assert p.white_point.tuple() == pytest.approx((1.18038779, 0.929053056, 0.890559155))


def test_transmittance_table_against_float64_quadrature(sky):
    """Transmittance to the top of the atmosphere for a few (r, mu): independent numpy integration."""
    p, luts = sky
    T = luts["transmittance"]
    assert T.shape == (64, 256, 4) and np.isfinite(T).all()
    assert (T[..., :3] > 0).all() and (T[..., :3] <= 1.0 + 1e-6).all()
    bot, top = 6360000.0, 6420000.0
    H = np.sqrt(top * top - bot * bot)
    bR = np.array(p.rayleigh_scattering.tuple()); bM = np.array(p.mie_extinction.tuple()); bO = np.array(p.absorption_extinction.tuple())

    def dens_ozone(h):
        return np.clip(np.where(h < 25000.0, h / 15000.0 - 2.0 / 3.0, -h / 15000.0 + 8.0 / 3.0), 0, 1)

    for (ix, iy) in ((255, 0), (200, 10), (128, 32), (40, 63), (250, 60)):
        u, v = (ix + 0.5) / 256, (iy + 0.5) / 64
        x_mu = (u - 0.5 / 256) / (1 - 1 / 256); x_r = (v - 0.5 / 64) / (1 - 1 / 64)
        rho = H * x_r
        r = np.sqrt(rho * rho + bot * bot)
        d_min, d_max = top - r, rho + H
        d = d_min + x_mu * (d_max - d_min)
        mu = 1.0 if d == 0 else np.clip((H * H - rho * rho - d * d) / (2 * r * d), -1, 1)
        disc = r * r * (mu * mu - 1) + top * top
        L = max(0.0, -r * mu + np.sqrt(max(disc, 0.0)))
        s = np.linspace(0, L, 20001)
        h = np.sqrt(s * s + 2 * r * mu * s + r * r) - bot
        oR = np.trapezoid(np.clip(np.exp(-h / 8000.0), 0, 1), s)
        oM = np.trapezoid(np.clip(np.exp(-h / 1200.0), 0, 1), s)
        oO = np.trapezoid(dens_ozone(h), s)
        exp = np.exp(-(bR * oR + bM * oM + bO * oO))
        np.testing.assert_allclose(T[iy, ix, :3], exp, rtol=2e-3, atol=1e-5)


def test_scattering_tables_are_sane(sky):
    p, luts = sky
    for name in ("irradiance", "scattering", "single_mie"):
        a = luts[name]
        assert np.isfinite(a).all(), name
        assert (a[..., :3] >= -1e-12).all(), name
    assert luts["single_mie"][..., :3].max() > 0 and luts["scattering"][..., :3].max() > 0 and luts["irradiance"][..., :3].max() > 0
    # reference quirk D3: the last multiple-scattering order overwrites the table, w = 0
    assert (luts["scattering"][..., 3] == 0).all()


def test_config2_sun_and_sky_parity(pkg, sky):
    """BASELINE config 2 at reduced size: sun NEE + procedural sky tail, HIP vs oracle, same tables."""
    import oracle_binding
    sd = pkg.scene.dragon_scene(160, 90, "c2")
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(3)
    hb.sync()
    ob = oracle_binding.OracleBinding(sd)
    ob.render(3)
    got = hb.accum.cpu().numpy()
    assert np.isfinite(got).all() and got.min() >= 0
    assert ob.accum.mean() > 1e-3                      # the sky actually contributes
    e = rel_l2(got, ob.accum)
    assert e <= 1e-3, e                                # north-star tolerance
    # achieved: the sky tail is value-only code built with approximate fp32 divide/sqrt and the hardware
    # exp/log (csrc/vpt_tail.hip); its ill-conditioned geometry terms amplify those ulps to ~1e-4
    assert e <= 4e-4, e
    np.testing.assert_allclose(hb.raw.cpu().numpy()[:, :3], ob.raw[:, :3], rtol=2e-3, atol=6e-4)


def test_camera_point_table_fast_path_matches_general_path(pkg, sky, monkeypatch):
    """tail_resolve's camera-point scattering table (csrc/vpt_sky.h: the 4-D tables pre-interpolated at
    the view point's r and mu_s) against the general 16-texel look-up, and the fall-back when the view
    point varies per sample (aperture > 0)."""
    sd = pkg.scene.dragon_scene(128, 72, "c2")
    sd.kp.sun_mult = 0.0                                  # sky only: the table's contribution is all there is
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    monkeypatch.setenv("VPT_NO_DIR_TABLE", "1")           # the ground table has its own test below
    monkeypatch.setenv("VPT_NO_SKY_PATCH", "1")           # ... and so have the per-pixel sky patches
    monkeypatch.setenv("VPT_NO_SKY_DOME", "1")            # ... and the sky domes (closed and open lens)
    fast = pkg.scene.HipBinding(sd, device=0)
    fast.render(2); fast.sync()
    monkeypatch.setenv("VPT_NO_CAM_TABLE", "1")
    slow = pkg.scene.HipBinding(sd, device=0)
    slow.render(2); slow.sync()
    monkeypatch.delenv("VPT_NO_CAM_TABLE")
    a, b = fast.accum.cpu().numpy(), slow.accum.cpu().numpy()
    assert b.mean() > 1e-2 and not np.array_equal(a, b)   # a different evaluation order ...
    assert rel_l2(a, b) <= 2e-6                           # ... of the same multilinear interpolant
    # aperture > 0: every sample starts somewhere on the lens disc, whose height spans a few binary32 steps of r: one table
    # variant per step (vpt_sky.h CamVariant) -- still the same multilinear interpolant as the general look-up
    cam, _, _ = pkg.scene.frame_camera(pkg.load_library(), [sd.volumes[0][0]], 128, 72, aperture=2.0)
    sd.camera = cam
    x = pkg.scene.HipBinding(sd, device=0)
    x.render(2); x.sync()
    monkeypatch.setenv("VPT_NO_CAM_TABLE", "1")
    y = pkg.scene.HipBinding(sd, device=0)
    y.render(2); y.sync()
    a, b = x.accum.cpu().numpy(), y.accum.cpu().numpy()
    assert not np.array_equal(a, b) and rel_l2(a, b) <= 2e-6, rel_l2(a, b)
    # a lens wider than the variants cover (4 steps = 2 m of height): samples beyond them take the general path, the rest the tables
    cam, _, _ = pkg.scene.frame_camera(pkg.load_library(), [sd.volumes[0][0]], 128, 72, aperture=12.0)
    sd.camera = cam
    monkeypatch.delenv("VPT_NO_CAM_TABLE")
    x = pkg.scene.HipBinding(sd, device=0)
    x.render(2); x.sync()
    monkeypatch.setenv("VPT_NO_CAM_TABLE", "1")
    y = pkg.scene.HipBinding(sd, device=0)
    y.render(2); y.sync()
    assert rel_l2(x.accum.cpu().numpy(), y.accum.cpu().numpy()) <= 2e-6
    import oracle_binding
    ob = oracle_binding.OracleBinding(sd)
    ob.render(2)
    assert rel_l2(x.accum.cpu().numpy(), ob.accum) <= 1e-3


def _dir_table_error(pkg, hb):
    import ctypes as C
    lib = pkg.load_library()
    lib.vpt_test_get_dir_table_error.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_uint)]
    built, err, cell = C.c_int(0), C.c_float(0), C.c_uint(0)
    assert lib.vpt_test_get_dir_table_error(hb.ctx.h, C.byref(built), C.byref(err), C.byref(cell)) == 0
    return built.value, err.value


@pytest.mark.parametrize("view", ["c2", "low sun", "sunset", "20 km up", "horizon in view"])
def test_view_point_ground_table_matches_full_evaluation(pkg, sky, monkeypatch, view):
    """tail_resolve's view-point ground table (csrc/vpt_sky.h GroundNode: the ground radiance, transmittance and ground-point
    scattering of a ray that ends on the ground, tabulated over distance x nu for the frame's view point and sun) against the
    full evaluation of every ground hit: its self-measured interpolation error is below the acceptance threshold, the images
    agree to 2e-4 relative L2 (the tolerance of the path is 1e-3), and both agree with the oracle; a threshold below the
    measured error switches the table off (bit-identical to VPT_NO_DIR_TABLE)."""
    import oracle_binding
    def make():
        sd = pkg.scene.dragon_scene(160, 90, "c2")
        if view == "low sun": sd.kp.elevation = 3.0
        if view == "sunset": sd.kp.elevation = -1.0
        if view == "20 km up": sd.camera.origin.y += 20000.0
        if view == "horizon in view":
            # a level, wide view from 3 m above the ground: the horizon crosses the image, so the grazing rays the table leaves to
            # the full path, the table's far end and the sky branch all take part
            import ctypes as C
            from vpt_amd.abi import Float3
            pkg.load_library().vpt_camera_update(C.byref(sd.camera), Float3(40.0, 3.0, 5.0), Float3(0.0, 3.0, 0.0), Float3(0, 1, 0), 70.0, 160.0 / 90.0, 0.0)
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
        return sd
    sd = make()
    tab = pkg.scene.HipBinding(sd, device=0)
    tab.render(4); tab.sync()
    built, err = _dir_table_error(pkg, tab)
    assert built == 1 and 0.0 < err <= 5e-4, err
    monkeypatch.setenv("VPT_NO_DIR_TABLE", "1")
    full = pkg.scene.HipBinding(sd, device=0)
    full.render(4); full.sync()
    assert _dir_table_error(pkg, full)[0] == 0
    monkeypatch.delenv("VPT_NO_DIR_TABLE")
    a, b = tab.accum.cpu().numpy(), full.accum.cpu().numpy()
    assert b.mean() > 1e-2 and not np.array_equal(a, b)      # rays do end on the ground, and the table is what evaluated them
    assert rel_l2(a, b) <= 2e-4, rel_l2(a, b)
    np.testing.assert_array_equal(tab.depth.cpu().numpy(), full.depth.cpu().numpy())
    ob = oracle_binding.OracleBinding(sd)
    ob.render(4)
    assert rel_l2(a, ob.accum) <= 4e-4 and rel_l2(b, ob.accum) <= 4e-4, (rel_l2(a, ob.accum), rel_l2(b, ob.accum))
    monkeypatch.setenv("VPT_DIR_TABLE_TOL", "1e-9")
    off = pkg.scene.HipBinding(sd, device=0)
    off.render(4); off.sync()
    np.testing.assert_array_equal(off.accum.cpu().numpy(), b)


def test_ground_table_behind_an_open_lens(pkg, sky, monkeypatch):
    """aperture > 0: the samples start on the lens disc; each binary32 value of r the disc reaches has its own table variant.  The
    full path's binary32 ground-point radius flips more often for such origins (vpt_sky.h), so the two evaluations agree to 4e-4
    rather than 2e-5 -- and both with the oracle"""
    import oracle_binding
    sd = pkg.scene.dragon_scene(128, 72, "c2")
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    sd.camera, _, _ = pkg.scene.frame_camera(pkg.load_library(), [sd.volumes[0][0]], 128, 72, aperture=2.0)
    x = pkg.scene.HipBinding(sd, device=0)
    x.render(4); x.sync()
    built, err = _dir_table_error(pkg, x)
    assert built == 1 and 0.0 < err <= 5e-4, err
    monkeypatch.setenv("VPT_NO_DIR_TABLE", "1")
    y = pkg.scene.HipBinding(sd, device=0)
    y.render(4); y.sync()
    a, b = x.accum.cpu().numpy(), y.accum.cpu().numpy()
    assert b.mean() > 1e-2 and not np.array_equal(a, b)
    assert rel_l2(a, b) <= 4e-4, rel_l2(a, b)
    np.testing.assert_array_equal(x.depth.cpu().numpy(), y.depth.cpu().numpy())
    ob = oracle_binding.OracleBinding(sd)
    ob.render(4)
    assert rel_l2(a, ob.accum) <= 5e-4 and rel_l2(b, ob.accum) <= 5e-4, (rel_l2(a, ob.accum), rel_l2(b, ob.accum))


def test_ground_table_per_direction(pkg, sky):
    """the same comparison per DIRECTION (vpt_test_sky_samples: sample_atmosphere from the view point of the last render): over the
    lower hemisphere, log-uniform in the angle below the horizontal, the table path equals the full path bit for bit where the table is
    not used (grazing rays, within ~2 degrees of the horizon) and agrees to 1e-4 for 99 % of the steeper directions; the rest are
    the ground points whose binary32 radius the full path finds one step above the ground (vpt_sky.h): under 1 % of them differ by
    more than 1e-3, none by more than 1e-2 -- the same figures the table's build-time check measures along its own real rays and
    gates the table on (sky_dir_table_rays_kernel: worst ray <= 2e-2, at most 2 % above 1e-3)"""
    import ctypes as C
    lib = pkg.load_library()
    lib.vpt_test_sky_samples.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    sd = pkg.scene.dragon_scene(64, 36, "c2")
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(1); hb.sync()
    assert _dir_table_error(pkg, hb)[0] == 1
    rng = np.random.default_rng(5)
    n = 1 << 18
    el = -np.exp(rng.uniform(np.log(1e-5), np.log(np.pi / 2), n))
    az = rng.uniform(0, 2 * np.pi, n)
    d = np.stack([np.cos(el) * np.cos(az), np.sin(el), np.cos(el) * np.sin(az)], 1).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    out = {}
    for use in (1, 0):
        o = np.zeros((n, 3), np.float32)
        assert lib.vpt_test_sky_samples(hb.ctx.h, n, None, d.ctypes.data, use, o.ctypes.data) == 0
        out[use] = o.astype(np.float64)
    assert np.isfinite(out[0]).all() and out[0].min() >= 0 and out[0].max() > 0.05
    rel = np.abs(out[1] - out[0]).max(1) / np.maximum(out[0].max(1), 1e-9)
    grazing = el > -0.02
    steep = el < -0.05
    assert (rel[grazing] == 0).all()
    assert (rel[steep] > 0).mean() > 0.3                      # the table is what evaluated them
    # Two populations (round 5, measured: profiles/r05_run1: median 3e-6, 97th percentile 4e-5, 99th 4.7e-3): the bulk follows the full path to
    # 1e-4; the rest are the FLIPPED rays -- the full path (now formed as the strict side forms it, so it lands where the oracle lands) finds their
    # binary32 ground point one step above the ground, ~2 % of the rays, ~5e-3 of the radiance each: noise of the reference that a smooth table returns
    # the mean of.  The build-time gate counts the two apart (vpt_tail.hip: sky_dir_table_rays_kernel).
    assert np.quantile(rel[steep], 0.96) <= 1e-4, np.quantile(rel[steep], [0.5, 0.96, 0.99])
    assert (rel[steep] > 1e-3).mean() <= 0.04, (rel[steep] > 1e-3).mean()
    assert rel.max() <= 1e-2, rel.max()
    # ... and the table's build-time check saw the same: real rays through both paths, worst ray, unflipped rays above 1e-3 and flipped rays within the gate
    chk = (C.c_float * 8)()
    lib.vpt_test_get_dir_table_check.argtypes = [C.c_void_p, C.POINTER(C.c_float * 8)]
    assert lib.vpt_test_get_dir_table_check(hb.ctx.h, C.byref(chk)) == 0
    assert chk[4] == 1.0 and chk[5] == 1.0 and chk[2] > 5000
    assert 0.0 < chk[1] <= 2e-2 and chk[3] <= 0.005 * chk[2], list(chk)
    flips = (C.c_float * 4)()
    lib.vpt_test_get_dir_table_flips.argtypes = [C.c_void_p, C.POINTER(C.c_float * 4)]
    assert lib.vpt_test_get_dir_table_flips(hb.ctx.h, C.byref(flips)) == 0
    assert 0.0 < flips[0] <= 0.06 and 0.0 < flips[2] <= 3e-4 and flips[3] <= 3e-4, list(flips)      # share of flipped rays, and what they cost on average


def test_ground_table_with_a_luminance_sky_model(pkg, monkeypatch):
    """the same with a sky model other than the defaults (vpt_atmosphere_model: luminance APPROXIMATE, no ozone, constant solar
    spectrum): the luminance factors multiply inside the table; table vs full evaluation, and both vs the oracle"""
    import oracle_binding
    sd = pkg.scene.dragon_scene(160, 90, "c2")
    pkg.atmosphere.attach_default_atmosphere(sd, device=0, use_luminance=1, use_ozone=0, use_constant_solar_spectrum=1)
    assert sd.atmosphere.use_luminance == 1
    tab = pkg.scene.HipBinding(sd, device=0)
    tab.render(4); tab.sync()
    built, err = _dir_table_error(pkg, tab)
    assert built == 1 and 0.0 < err <= 5e-4, err
    monkeypatch.setenv("VPT_NO_DIR_TABLE", "1")
    full = pkg.scene.HipBinding(sd, device=0)
    full.render(4); full.sync()
    a, b = tab.accum.cpu().numpy(), full.accum.cpu().numpy()
    assert b.mean() > 1e-2 and not np.array_equal(a, b)
    assert rel_l2(a, b) <= 2e-4, rel_l2(a, b)
    ob = oracle_binding.OracleBinding(sd)
    ob.render(4)
    assert rel_l2(a, ob.accum) <= 1e-3 and rel_l2(b, ob.accum) <= 1e-3, (rel_l2(a, ob.accum), rel_l2(b, ob.accum))


def test_per_frame_sky_tables_follow_the_lut_contents(pkg):
    """The camera-point and ground tables of the environment tail are cached on the view point, the sun, the model SCALARS and the
    ADDRESSES of the four atmosphere tables (round-2 advisor finding: not on their contents).  Two skies with identical scalars and
    different tables (ozone on / off changes only the absorption profile, which is not among the packed scalars):
      * tables re-uploaded through destroy + create (the new allocation usually lands on the address just freed): the cache is
        dropped by the texture calls themselves;
      * a device table the host owns, rewritten IN PLACE: vpt_invalidate_sky_tables.
    Either way the next render must equal a fresh context's, bit for bit."""
    import torch
    from vpt_amd import abi
    wc = (abi.ADDR_WRAP, abi.ADDR_CLAMP, abi.ADDR_CLAMP)
    sd = pkg.scene.dragon_scene(128, 72, "c2")
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    sb = pkg.scene.dragon_scene(128, 72, "c2")
    pkg.atmosphere.attach_default_atmosphere(sb, device=0, use_ozone=0)
    la, lb = sd.atm_luts, sb.atm_luts
    assert not np.array_equal(la["scattering"], lb["scattering"])
    # reference: a fresh context with A's scalars and B's tables
    sc = pkg.scene.dragon_scene(128, 72, "c2")
    sc.atmosphere = abi.AtmosphereParameters.from_buffer_copy(sd.atmosphere)
    sc.atm_luts = lb
    hc = pkg.scene.HipBinding(sc, device=0)
    hc.render(2, iteration=0); hc.sync()
    want = hc.accum.cpu().numpy()

    def bind(hb, make):
        hb.atmosphere.transmittance_texture = make("transmittance", wc)
        hb.atmosphere.irradiance_texture = make("irradiance", wc)
        hb.atmosphere.scattering_texture = make("scattering", (abi.ADDR_CLAMP,) * 3)
        hb.atmosphere.single_mie_scattering_texture = make("single_mie", (abi.ADDR_CLAMP,) * 3)

    # (1) destroy + create
    hb = pkg.scene.HipBinding(sd, device=0)
    bn0 = hb.blue_noise.clone()
    hb.render(2, iteration=0); hb.sync()
    first = hb.accum.cpu().numpy().copy()
    assert not np.array_equal(first, want)
    for t in (hb.atmosphere.transmittance_texture, hb.atmosphere.irradiance_texture, hb.atmosphere.scattering_texture,
              hb.atmosphere.single_mie_scattering_texture):
        assert hb.ctx.lib.vpt_texture_destroy(hb.ctx.h, t) == 0
    bind(hb, lambda name, addr: hb.ctx.texture(lb[name], 4, address=addr))
    hb.blue_noise.copy_(bn0); torch.cuda.synchronize()
    hb.render(2, iteration=0); hb.sync()
    np.testing.assert_array_equal(hb.accum.cpu().numpy(), want)

    # (2) device tables rewritten in place
    dev = {k: torch.from_numpy(la[k].copy()).to("cuda:0") for k in la}
    hd = pkg.scene.HipBinding(sd, device=0)
    dims = {"transmittance": (256, 64, 1), "irradiance": (256, 64, 1), "scattering": (256, 128, 32), "single_mie": (256, 128, 32)}
    bind(hd, lambda name, addr: hd.ctx.texture_device(dev[name], dims[name], 4, address=addr))
    torch.cuda.synchronize()
    hd.render(2, iteration=0); hd.sync()
    np.testing.assert_array_equal(hd.accum.cpu().numpy(), first)
    for k in dev:
        dev[k].copy_(torch.from_numpy(lb[k]))
    torch.cuda.synchronize()
    assert hd.ctx.lib.vpt_invalidate_sky_tables(hd.ctx.h) == 0
    hd.blue_noise.copy_(bn0); torch.cuda.synchronize()
    hd.render(2, iteration=0); hd.sync()
    np.testing.assert_array_equal(hd.accum.cpu().numpy(), want)


@pytest.mark.parametrize("view", ["default", "horizon in view", "sun in view", "1080p"])
def test_sky_patch_matches_full_evaluation(pkg, sky, monkeypatch, view):
    """Untraced samples behind a closed lens take their sky value from a per-pixel bilinear patch over the sample's jitter
    (csrc/vpt_tail.hip: sky_patch_kernel -- four corner evaluations per pixel, kept only where the patch reproduces the exact centre
    value to 1e-3 and the sun's disc is not near) instead of evaluating sample_atmosphere per sample.  Against VPT_NO_SKY_PATCH=1
    (every sample in full): images within 3e-4 relative L2 at 160 x 90 and 1e-4 at 1080p, no pixel off by more than 2e-3 of its own brightness, depth and alpha
    bit-identical, and both within the path's tolerance of the oracle -- also with the horizon or the sun's disc in the picture,
    where the patch must step aside."""
    import ctypes as C
    import oracle_binding
    from vpt_amd.abi import Float3
    w, h = (1920, 1080) if view == "1080p" else (160, 90)
    sd = pkg.scene.dragon_scene(w, h, "c2")
    lib = pkg.load_library()
    if view == "horizon in view":
        lib.vpt_camera_update(C.byref(sd.camera), Float3(40.0, 3.0, 5.0), Float3(0.0, 3.0, 0.0), Float3(0, 1, 0), 70.0, w / h, 0.0)
    if view == "sun in view":
        # look from the volume towards the sun
        az, el = np.radians(sd.kp.azimuth), np.radians(90.0 - sd.kp.elevation)            # degree_to_cartesian (render_kernel.cu:126-142)
        s = np.array([np.sin(el) * np.cos(az), np.cos(el), np.sin(el) * np.sin(az)])
        o = np.array([sd.camera.origin.x, sd.camera.origin.y, sd.camera.origin.z])
        t = o + 100.0 * s
        lib.vpt_camera_update(C.byref(sd.camera), Float3(*o), Float3(*t), Float3(0, 1, 0), 40.0, w / h, 0.0)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    n = 2 if view == "1080p" else 4
    monkeypatch.setenv("VPT_NO_SKY_DOME", "1")                 # (the dome has its own test below)
    a = pkg.scene.HipBinding(sd, device=0)
    a.render(n); a.sync()
    monkeypatch.setenv("VPT_NO_SKY_PATCH", "1")
    b = pkg.scene.HipBinding(sd, device=0)
    b.render(n); b.sync()
    monkeypatch.delenv("VPT_NO_SKY_PATCH")
    x, y = a.accum.cpu().numpy().astype(np.float64), b.accum.cpu().numpy().astype(np.float64)
    assert y.mean() > 1e-3 and not np.array_equal(x, y)                   # the patch is what evaluated (most of) the untraced samples
    assert rel_l2(x, y) <= (1e-4 if view == "1080p" else 3e-4), rel_l2(x, y)      # (160 x 90: pixels of 0.4 degrees; 1080p: 0.03)
    lum = y.max(1)
    rel = np.abs(x - y).max(1)[lum > 1e-3] / lum[lum > 1e-3]
    # per pixel: 2e-3 of the pixel's brightness -- except in the few degrees of SKY above the horizon, where sample_atmosphere itself
    # jumps by ~1 % from ray to ray (the reference's binary32 r^2 mu^2 - r^2 + bottom^2 next to mu = 0): there the means of four
    # noisy samples and the smooth patch differ by up to 8e-3, whatever the resolution (measured: 0.2 % of a 1080p frame's pixels
    # above 2e-3 with the horizon across it, none without)
    assert rel.max() <= (1e-2 if view == "horizon in view" else 2e-3), rel.max()
    assert (rel > 2e-3).mean() <= 0.01
    cov = (C.c_ulonglong(0), C.c_ulonglong(0))
    lib.vpt_test_get_sky_patch_coverage.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    assert lib.vpt_test_get_sky_patch_coverage(a.ctx.h, C.byref(cov[0]), C.byref(cov[1])) == 0
    assert cov[0].value == w * h and cov[1].value >= (0.5 if view == "horizon in view" else 0.9) * w * h, (cov[0].value, cov[1].value)
    assert cov[1].value < w * h or view in ("default", "1080p")            # the horizon / the sun's disc switch some pixels' patches off
    np.testing.assert_array_equal(a.depth.cpu().numpy(), b.depth.cpu().numpy())
    np.testing.assert_array_equal(a.raw.cpu().numpy()[:, 3], b.raw.cpu().numpy()[:, 3])
    if view != "1080p":
        ob = oracle_binding.OracleBinding(sd)
        ob.render(n)
        assert rel_l2(x, ob.accum) <= 5e-4 and rel_l2(y, ob.accum) <= 5e-4, (rel_l2(x, ob.accum), rel_l2(y, ob.accum))


@pytest.mark.parametrize("view", ["default", "horizon in view", "sphere in view", "inside the box", "render off"])
def test_never_traced_pixels_change_nothing(pkg, sky, monkeypatch, view):
    """Pixels whose whole jitter footprint misses the volumes' root box and the reference sphere -- screen-space bounds grown by three
    pixels, the `B == 0` line of sphere::intersect kept out (csrc/vpt_host.hip: project_box, vpt_tail.hip: sky_patch_kernel) -- and that
    have a sky patch are skipped by raygen altogether; the tail takes their samples from the patch.  Against VPT_NO_PIXEL_CULL=1:
    every buffer and every count bit-identical, and some pixels really were skipped (except from inside the box)."""
    import ctypes as C
    from vpt_amd.abi import Float3
    w, h = 320, 180
    sd = pkg.scene.dragon_scene(w, h, "c2")
    lib = pkg.load_library()
    if view == "horizon in view":
        lib.vpt_camera_update(C.byref(sd.camera), Float3(40.0, 3.0, 5.0), Float3(0.0, 3.0, 0.0), Float3(0, 1, 0), 70.0, w / h, 0.0)
    if view == "sphere in view":
        o = sd.camera.origin
        sd.sphere.center = Float3(o.x * 0.55, o.y * 0.55 + 1.0, o.z * 0.55 - 2.0)
        sd.sphere.radius = 1.5
    if view == "inside the box":
        lib.vpt_camera_update(C.byref(sd.camera), Float3(0.3, 0.2, 0.1), Float3(5.0, 1.0, 2.0), Float3(0, 1, 0), 60.0, w / h, 0.0)
    if view == "render off":
        sd.kp.max_interactions = 2                       # iterations 2.. are not rendered (volume_rt_kernel :2254)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)

    def run():
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.ctx.set_counting(True)
        hb.render(4); hb.sync()
        return hb, hb.ctx.stats()
    a, sa = run()
    monkeypatch.setenv("VPT_NO_PIXEL_CULL", "1")
    b, sb = run()
    for buf in ("accum", "depth", "raw", "display"):
        np.testing.assert_array_equal(getattr(a, buf).cpu().numpy(), getattr(b, buf).cpu().numpy(), err_msg=buf)
    for k in ("samples", "density_lookups", "tracking_steps", "skip_steps", "queued_rays"):
        assert getattr(sa, k) == getattr(sb, k), k
    assert sa.samples == w * h * 4


@pytest.mark.parametrize("scene", ["dragon", "fireball", "instances"])
def test_never_traced_masks_on_random_views(pkg, sky, monkeypatch, scene):
    """The two tests around this one compare hand-picked views; this one sweeps SEEDED RANDOM cameras -- near and far, from below the horizon to the zenith, narrow and
    wide fields of view, looking at the volume or past it, with and without the reference sphere in view, odd image extents -- over three scenes (one grid with empty
    octree nodes, one without, an instanced one).  For every view: the timed render with both masks (root-box bounds + leaf tiles) against VPT_NO_PIXEL_CULL=1 (raygen
    emits every sample), every buffer bit-identical and the same rays queued.  The masks hold "by test": this is the widest net the suite casts."""
    import ctypes as C
    from vpt_amd.abi import Float3
    lib = pkg.load_library()
    lib.vpt_test_count_never_traced.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    rs = np.random.RandomState({"dragon": 11, "fireball": 12, "instances": 13}[scene])
    skipped_views = traced_views = 0
    for view in range(10):
        w, h = int(rs.choice([320, 333, 256])), int(rs.choice([180, 187, 144]))
        if scene == "dragon":
            sd = pkg.scene.dragon_scene(w, h, "c2")
        elif scene == "fireball":
            sd = pkg.scene.fireball_scene(w, h, n=48, sky=True)
        else:
            sd = pkg.scene.instanced_scene(w, h, n=32, grid=3, aperture=0.0, sky=True)
        o = sd.camera.origin
        dist = float(np.sqrt(o.x * o.x + o.y * o.y + o.z * o.z)) * float(rs.uniform(0.35, 2.5))
        az, el = float(rs.uniform(0.0, 2.0 * np.pi)), float(rs.uniform(-0.15, 1.45))
        eye = (dist * np.cos(el) * np.cos(az), dist * np.sin(el) + float(rs.uniform(0.0, 2.0)), dist * np.cos(el) * np.sin(az))
        look = tuple(float(v) for v in rs.uniform(-1.0, 1.0, 3) * (dist * 0.3 if view % 3 == 2 else 1.0))       # every third view looks well past the volume
        lib.vpt_camera_update(C.byref(sd.camera), Float3(*eye), Float3(*look), Float3(0, 1, 0), float(rs.uniform(18.0, 85.0)), w / h, 0.0)
        if view % 4 == 1:                                  # the reference sphere somewhere between the camera and the volume
            t = float(rs.uniform(0.3, 0.7))
            sd.sphere.center = Float3(eye[0] * t + float(rs.uniform(-1, 1)), eye[1] * t + float(rs.uniform(-1, 1)), eye[2] * t + float(rs.uniform(-1, 1)))
            sd.sphere.radius = float(rs.uniform(0.3, 1.5))
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)

        def run():
            hb = pkg.scene.HipBinding(sd, device=0)
            hb.render(3); hb.sync()
            n = C.c_ulonglong(0)
            assert lib.vpt_test_count_never_traced(hb.ctx.h, C.byref(n)) == 0
            out = {b: getattr(hb, b).cpu().numpy().copy() for b in ("accum", "depth", "raw", "display")}
            st = hb.ctx.stats()
            hb.ctx.close()
            return out, st, n.value
        a, sa, na = run()
        monkeypatch.setenv("VPT_NO_PIXEL_CULL", "1")
        b, sb, nb = run()
        monkeypatch.delenv("VPT_NO_PIXEL_CULL")
        assert nb == 0
        for buf in a:
            np.testing.assert_array_equal(a[buf], b[buf], err_msg="%s view %d: %s" % (scene, view, buf))
        assert sa.queued_rays == sb.queued_rays, (scene, view)
        assert np.isfinite(a["accum"]).all()
        skipped_views += na > 0
        traced_views += sa.queued_rays > 0
        print("%s view %d (%d x %d): %d of %d pixels never traced, %d rays queued" % (scene, view, w, h, na, w * h, sa.queued_rays))
    assert skipped_views >= 5 and traced_views >= 5, (skipped_views, traced_views)            # the sweep really exercises the masks, on views that see the volume


@pytest.mark.parametrize("view", ["default", "1080p", "horizon in view", "sphere in view", "inside the box"])
def test_leaf_level_never_traced_tiles_change_nothing(pkg, sky, monkeypatch, view):
    """The never-traced mask refined per 8x8-pixel tile by the screen bounds of the NON-EMPTY octree leaves (ResolveParams::cull_tiles): a ray that only
    ever crosses empty nodes is pushed out of the root without a draw or a look-up (render_kernel.cu:1606-1616) and, with nothing behind, ends exactly
    as a ray that misses the box -- raygen walks those pushes per sample; a tile no non-empty leaf can be seen through needs none of it.  Against
    VPT_NO_LEAF_CULL=1 (the mask from the root box's bounds alone): every buffer bit-identical, the same rays queued, and more pixels skipped.
    (Counting renders do not refine: they report the reference-defined skip counts of exactly those rays.)"""
    import ctypes as C
    from vpt_amd.abi import Float3
    w, h = (1920, 1080) if view == "1080p" else (320, 180)
    sd = pkg.scene.dragon_scene(w, h, "c2")
    lib = pkg.load_library()
    if view == "horizon in view":
        lib.vpt_camera_update(C.byref(sd.camera), Float3(40.0, 3.0, 5.0), Float3(0.0, 3.0, 0.0), Float3(0, 1, 0), 70.0, w / h, 0.0)
    if view == "sphere in view":
        o = sd.camera.origin
        sd.sphere.center = Float3(o.x * 0.55, o.y * 0.55 + 1.0, o.z * 0.55 - 2.0)
        sd.sphere.radius = 1.5
    if view == "inside the box":
        lib.vpt_camera_update(C.byref(sd.camera), Float3(0.3, 0.2, 0.1), Float3(5.0, 1.0, 2.0), Float3(0, 1, 0), 60.0, w / h, 0.0)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    lib.vpt_test_get_cache_state.argtypes = [C.c_void_p, C.POINTER(C.c_int * 8)]
    lib.vpt_test_count_never_traced.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]

    def run():
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.render(4); hb.sync()
        st = (C.c_int * 8)()
        assert lib.vpt_test_get_cache_state(hb.ctx.h, C.byref(st)) == 0
        n = C.c_ulonglong(0)
        assert lib.vpt_test_count_never_traced(hb.ctx.h, C.byref(n)) == 0
        return hb, hb.ctx.stats(), list(st), n.value
    a, sa, ca, na = run()
    monkeypatch.setenv("VPT_NO_LEAF_CULL", "1")
    b, sb, cb, nb = run()
    for buf in ("accum", "depth", "raw", "display"):
        np.testing.assert_array_equal(getattr(a, buf).cpu().numpy(), getattr(b, buf).cpu().numpy(), err_msg=buf)
    assert sa.queued_rays == sb.queued_rays and sa.queued_rays > 0
    assert cb[7] == 0
    assert ca[1] == 1 and ca[7] == 1, ca
    assert na >= nb > 0, (na, nb)
    if view in ("default", "1080p", "sphere in view"):
        assert na > 1.2 * nb, (na, nb)                        # the dragon fills a fraction of its padded box
    print("never-traced pixels (%s): %d of %d with the leaf tiles, %d from the root box alone" % (view, na, w * h, nb))
    # ... and a counting render keeps the root-box mask (and with it the skip counts the oracle reports)
    monkeypatch.delenv("VPT_NO_LEAF_CULL")
    hc = pkg.scene.HipBinding(sd, device=0)
    hc.ctx.set_counting(True)
    hc.render(4); hc.sync()
    st = (C.c_int * 8)()
    assert lib.vpt_test_get_cache_state(hc.ctx.h, C.byref(st)) == 0 and st[7] == 0
    np.testing.assert_array_equal(hc.accum.cpu().numpy(), a.accum.cpu().numpy())


@pytest.mark.parametrize("view", ["default", "horizon in view", "sphere in view", "inside the box", "render off", "chunks"])
def test_resolved_samples_change_nothing(pkg, sky, monkeypatch, view):
    """With the per-view caches in use behind a closed lens the TRACER adds a finished path's environment term (a sky-dome look-up where ~44
    lanes finish together), sky_fix_kernel evaluates in full what neither the dome nor a patch serves (flagged dome cells, paths the sphere
    bounce moved, the pixels without a usable patch), and the tail streams 16-byte heads + 8-byte {alpha, depth} pairs several iterations
    ahead of its ordered running means (csrc/vpt_device.h: ResolveParams::lean).  Against VPT_NO_LEAN_TAIL=1 -- 64-byte path records, the
    environment added inside the tail's per-pixel loop: the same operations on the same values in the same order, so every buffer is
    bit-identical, on views that exercise every kind of sample."""
    import ctypes as C
    from vpt_amd.abi import Float3
    w, h = 320, 180
    sd = pkg.scene.dragon_scene(w, h, "c2")
    lib = pkg.load_library()
    if view == "horizon in view":
        lib.vpt_camera_update(C.byref(sd.camera), Float3(40.0, 3.0, 5.0), Float3(0.0, 3.0, 0.0), Float3(0, 1, 0), 70.0, w / h, 0.0)
    if view == "sphere in view":
        o = sd.camera.origin
        sd.sphere.center = Float3(o.x * 0.55, o.y * 0.55 + 1.0, o.z * 0.55 - 2.0)
        sd.sphere.radius = 1.5
    if view == "inside the box":
        lib.vpt_camera_update(C.byref(sd.camera), Float3(0.3, 0.2, 0.1), Float3(5.0, 1.0, 2.0), Float3(0, 1, 0), 60.0, w / h, 0.0)
    if view == "render off":
        sd.kp.max_interactions = 2
    if view == "chunks":
        monkeypatch.setenv("VPT_BATCH_ITERS", "3")       # 7 iterations in launches of 3 + 3 + 1
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    n = 7 if view == "chunks" else 4

    def run():
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.ctx.set_counting(True)
        hb.render(n); hb.sync()
        state = (C.c_int * 8)()
        lib.vpt_test_get_cache_state.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        assert lib.vpt_test_get_cache_state(hb.ctx.h, state) == 0
        return hb, hb.ctx.stats(), list(state)
    a, sa, ca = run()
    assert ca[0] and ca[2] and ca[6], ca                  # patches, dome, resolved samples
    monkeypatch.setenv("VPT_NO_LEAN_TAIL", "1")
    b, sb, cb = run()
    assert cb[0] and cb[2] and not cb[6], cb
    for buf in ("accum", "depth", "raw", "display"):
        np.testing.assert_array_equal(getattr(a, buf).cpu().numpy(), getattr(b, buf).cpu().numpy(), err_msg=buf)
    for k in ("samples", "density_lookups", "tracking_steps", "skip_steps", "queued_rays"):
        assert getattr(sa, k) == getattr(sb, k), k
    assert np.isfinite(a.accum.cpu().numpy()).all() and float(a.accum.max()) > 0


@pytest.mark.parametrize("view", ["default", "horizon in view", "sun in view", "1080p"])
def test_sky_dome_matches_full_evaluation(pkg, sky, monkeypatch, view):
    """Traced samples that look from the camera origin take the environment term from the SKY DOME (csrc/vpt_tail.hip: sky_dome_kernel -- the
    value of sample_atmosphere over the whole sphere of directions, 4096 x 2048 nodes, each cell kept only where its bilinear interpolant
    reproduces the exact centre value to 1e-3, holds no full-path ground hit and is away from the sun's disc) instead of evaluating it per
    sample.  Against VPT_NO_SKY_DOME=1: images within 1e-4 relative L2, pixels within 4e-3 of their brightness, 99.9 % of them within 2e-3 (1e-2 in the band of sky above
    the horizon where the per-sample evaluation itself scatters), depth and alpha bit-identical."""
    import ctypes as C
    from vpt_amd.abi import Float3
    w, h = (1920, 1080) if view == "1080p" else (320, 180)
    sd = pkg.scene.dragon_scene(w, h, "c2")
    lib = pkg.load_library()
    if view == "horizon in view":
        lib.vpt_camera_update(C.byref(sd.camera), Float3(40.0, 3.0, 5.0), Float3(0.0, 3.0, 0.0), Float3(0, 1, 0), 70.0, w / h, 0.0)
    if view == "sun in view":
        az, el = np.radians(sd.kp.azimuth), np.radians(90.0 - sd.kp.elevation)
        s = np.array([np.sin(el) * np.cos(az), np.cos(el), np.sin(el) * np.sin(az)])
        o = np.array([sd.camera.origin.x, sd.camera.origin.y, sd.camera.origin.z])
        lib.vpt_camera_update(C.byref(sd.camera), Float3(*o), Float3(*(o + 100.0 * s)), Float3(0, 1, 0), 40.0, w / h, 0.0)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    n = 2 if view == "1080p" else 4
    a = pkg.scene.HipBinding(sd, device=0)
    a.render(n); a.sync()
    monkeypatch.setenv("VPT_NO_SKY_DOME", "1")
    b = pkg.scene.HipBinding(sd, device=0)
    b.render(n); b.sync()
    x, y = a.accum.cpu().numpy().astype(np.float64), b.accum.cpu().numpy().astype(np.float64)
    assert y.mean() > 1e-3
    if view != "sun in view":                                 # (looking away from the volume nothing is traced: the dome has nothing to do)
        assert not np.array_equal(x, y)
    assert rel_l2(x, y) <= 1e-4, rel_l2(x, y)
    lum = y.max(1)
    rel = np.abs(x - y).max(1)[lum > 1e-3] / lum[lum > 1e-3]
    # (measured at 1080p: one pixel of the volume at 2.5e-3, the rest below 2e-3 -- a cell's gate is 1e-3 at its centre, and the per-sample
    # evaluation it is compared with carries the ~1e-4 staircase of the reference's binary32 scattering row)
    assert rel.max() <= (1e-2 if view == "horizon in view" else 4e-3), rel.max()
    assert (rel > 2e-3).mean() <= 0.001
    np.testing.assert_array_equal(a.depth.cpu().numpy(), b.depth.cpu().numpy())
    np.testing.assert_array_equal(a.raw.cpu().numpy()[:, 3], b.raw.cpu().numpy()[:, 3])


def test_sky_dome_behind_an_open_lens(pkg, sky, monkeypatch):
    """aperture > 0: every sample starts somewhere on the lens disc, and the sky sees that origin only through r and mu_s -- one dome per table
    variant (per binary32 value of r across the disc) serves untraced and traced samples alike.  Against VPT_NO_SKY_DOME=1: image within 2e-4
    relative L2, 99 % of the pixels within 2e-3, depth and alpha bit-identical; and both within the path's tolerance of the oracle."""
    import oracle_binding
    sd = pkg.scene.dragon_scene(320, 180, "c2")
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    sd.camera, _, _ = pkg.scene.frame_camera(pkg.load_library(), [sd.volumes[0][0]], 320, 180, aperture=2.0)
    a = pkg.scene.HipBinding(sd, device=0)
    a.render(4); a.sync()
    monkeypatch.setenv("VPT_NO_SKY_DOME", "1")
    b = pkg.scene.HipBinding(sd, device=0)
    b.render(4); b.sync()
    x, y = a.accum.cpu().numpy().astype(np.float64), b.accum.cpu().numpy().astype(np.float64)
    assert y.mean() > 1e-3 and not np.array_equal(x, y)
    assert rel_l2(x, y) <= 2e-4, rel_l2(x, y)
    lum = y.max(1)
    rel = np.abs(x - y).max(1)[lum > 1e-3] / lum[lum > 1e-3]
    assert (rel > 2e-3).mean() <= 0.01 and rel.max() <= 1e-2, ((rel > 2e-3).mean(), rel.max())
    np.testing.assert_array_equal(a.depth.cpu().numpy(), b.depth.cpu().numpy())
    np.testing.assert_array_equal(a.raw.cpu().numpy()[:, 3], b.raw.cpu().numpy()[:, 3])
    ob = oracle_binding.OracleBinding(sd)
    ob.render(4)
    assert rel_l2(x, ob.accum) <= 6e-4 and rel_l2(y, ob.accum) <= 6e-4, (rel_l2(x, ob.accum), rel_l2(y, ob.accum))
