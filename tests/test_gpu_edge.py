"""Edge cases of volume_rt_kernel's own bookkeeping (render_kernel.cu:2227-2326), HIP vs oracle:
frozen accumulation past max_interactions, render = false, viz_dof tint, exposure, the HDRI
background of direct_integrator, ragged resolutions, batches that span several record chunks."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(1e-30, np.sqrt((b ** 2).sum())))


def _both(pkg, sd, n, atmosphere=False, **render_kw):
    import oracle_binding
    if atmosphere:
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    hb = pkg.scene.HipBinding(sd, device=0)
    ob = oracle_binding.OracleBinding(sd)
    hb.render(n, **render_kw)
    hb.sync()
    ob.render(n, **render_kw)
    return hb, ob


def _same(hb, ob, tol=2e-6):
    got = hb.accum.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_l2(got, ob.accum) <= tol
    np.testing.assert_allclose(hb.depth.cpu().numpy(), ob.depth, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(hb.raw.cpu().numpy(), ob.raw, rtol=2e-4, atol=2e-5)
    disp = hb.display.cpu().numpy().view(np.uint32)
    sh = np.array([16, 8, 0])
    assert np.abs(((disp[:, None] >> sh) & 255).astype(int) - ((ob.display[:, None] >> sh) & 255).astype(int)).max() <= 1
    np.testing.assert_array_equal(hb.blue_noise.cpu().numpy(), ob.blue_noise)


def test_accumulation_freezes_past_max_interactions(pkg):
    """iteration >= max_interactions: the sample is WHITE and is not accumulated (:2254, :2282)"""
    sd = pkg.scene.dragon_scene(97, 61, "sun")               # ragged size: partial raygen tiles
    sd.kp.max_interactions = 3
    hb, ob = _both(pkg, sd, 6)
    _same(hb, ob)
    sd2 = pkg.scene.dragon_scene(97, 61, "sun")
    hb2, _ = _both(pkg, sd2, 3)
    np.testing.assert_array_equal(hb.accum.cpu().numpy(), hb2.accum.cpu().numpy())     # iterations 3..5 changed nothing


def test_render_false_yields_white(pkg):
    sd = pkg.scene.dragon_scene(64, 48, "c1")
    sd.kp.render = 0
    hb, ob = _both(pkg, sd, 2)
    _same(hb, ob)
    assert (hb.accum.cpu().numpy() == 1.0).all()             # value = WHITE (:2248)


def test_viz_dof_and_exposure(pkg):
    sd = pkg.scene.dragon_scene(96, 64, "sun")
    lib = pkg.load_library()
    import ctypes as C
    cam, _, _ = pkg.scene.frame_camera(lib, [sd.volumes[0][0]], 96, 64, aperture=0.8)
    cam.viz_dof = 1
    sd.camera = cam
    sd.kp.exposure_scale = 1.7
    hb, ob = _both(pkg, sd, 3)
    _same(hb, ob)
    a = hb.accum.cpu().numpy()
    assert (np.abs(a[:, 0] - a[:, 1]) > 1e-3).any()           # the tint is there


def test_direct_integrator_hdri_background(pkg):
    """environment_type 1 with integrator 0: lat-long look-up * sky_color * beta / 4 pi (:1843-1850)"""
    sd = pkg.scene.dragon_scene(128, 72, "c2")
    sd.kp.environment_type = 1
    sd.kp.sky_color = pkg.abi.Float3(0.9, 1.0, 1.1)
    sd.env_map = pkg.scene.hdri_map(256, 128)
    hb, ob = _both(pkg, sd, 3)
    got = hb.accum.cpu().numpy()
    assert rel_l2(got, ob.accum) <= 1e-3                      # atan2/acos are value-only arithmetic
    assert rel_l2(got, ob.accum) <= 2e-5
    assert got.mean() > 1e-2


def test_batches_spanning_record_chunks(pkg, monkeypatch):
    """a 7-iteration batch rendered in chunks of 2 iterations == the same batch in one chunk, starting
    at a non-zero iteration"""
    sd = pkg.scene.dragon_scene(80, 50, "sun")
    import oracle_binding
    monkeypatch.setenv("VPT_BATCH_ITERS", "2")
    hb = pkg.scene.HipBinding(sd, device=0)
    hb.render(7, iteration=0)
    hb.sync()
    monkeypatch.delenv("VPT_BATCH_ITERS")
    hb1 = pkg.scene.HipBinding(sd, device=0)
    hb1.render(7, iteration=0)
    hb1.sync()
    np.testing.assert_array_equal(hb.accum.cpu().numpy(), hb1.accum.cpu().numpy())
    np.testing.assert_array_equal(hb.display.cpu().numpy(), hb1.display.cpu().numpy())
    ob = oracle_binding.OracleBinding(sd)
    ob.render(7)
    assert rel_l2(hb.accum.cpu().numpy(), ob.accum) <= 2e-6


def test_nan_guard_substitutes_running_mean(pkg):
    """a NaN sample (here: a NaN albedo poisons every interacting path) is replaced by the current
    mean (:2263) -- pixels whose paths never interact are untouched"""
    sd = pkg.scene.dragon_scene(64, 48, "sun")
    hb0, _ = _both(pkg, sd, 1)
    base = hb0.accum.cpu().numpy().copy()
    import oracle_binding
    sd.kp.albedo = pkg.abi.Float3(float("nan"), 1.0, 1.0)
    hb = pkg.scene.HipBinding(sd, device=0)
    ob = oracle_binding.OracleBinding(sd)
    # iteration 0 clean (albedo patched after the first launch), iteration 1 poisoned
    clean = pkg.abi.Float3(1.0, 1.0, 1.0)
    hb.kp.albedo = clean; ob.kp.albedo = clean
    hb.render(1); hb.sync(); ob.render(1)
    hb.kp.albedo = sd.kp.albedo; ob.kp.albedo = sd.kp.albedo
    hb.render(1); hb.sync(); ob.render(1)
    got = hb.accum.cpu().numpy()
    assert np.isfinite(got).all() and np.isfinite(ob.accum).all()
    assert rel_l2(got, ob.accum) <= 2e-6


@pytest.mark.parametrize("layout", ["bricks", "quads"])
@pytest.mark.parametrize("scene", ["dragon", "fireball", "instanced", "cloud_vol"])
def test_relaid_density_layouts_are_bit_identical(pkg, monkeypatch, scene, layout):
    """large density grids are re-laid as float4 corner quads, or 4x4x4 bricks when those do not fit (data
    layout); forcing either on small scenes must not change a single bit (odd extents: 70x49x31 has partial edge bricks, and
    footprints on every face, edge and corner of the grid exercise the quads' clamped rows), and the result is the oracle's:
    bit-identical depth, image to 2e-6 (1e-3 where the value-only sky code takes part)"""
    import oracle_binding
    def make():
        if scene == "dragon":
            return pkg.scene.dragon_scene(96, 64, "sun")
        if scene == "fireball":
            return pkg.scene.fireball_scene(96, 64, n=37)
        if scene == "cloud_vol":
            sd = pkg.scene.cloud_scene(96, 64, shape=(45, 31, 38), env=(64, 32))        # vol_integrator: the split-phase look-up
            pkg.atmosphere.attach_default_atmosphere(sd, device=0)
            return sd
        return pkg.scene.instanced_scene(96, 64, n=18, grid=3, aperture=0.3)
    sd = make()
    a = pkg.scene.HipBinding(sd, device=0)
    a.render(3); a.sync()
    monkeypatch.setenv("VPT_RELAID_MIN_BYTES", "0")
    monkeypatch.setenv("VPT_GRID_LAYOUT", layout)
    b = pkg.scene.HipBinding(sd, device=0)
    b.ctx.set_counting(True)
    b.render(3); b.sync()
    assert a.accum.abs().max() > 0
    np.testing.assert_array_equal(a.accum.cpu().numpy(), b.accum.cpu().numpy())
    np.testing.assert_array_equal(a.depth.cpu().numpy(), b.depth.cpu().numpy())
    ob = oracle_binding.OracleBinding(sd)
    ob.render(3)
    np.testing.assert_array_equal(b.depth.cpu().numpy(), ob.depth)
    assert rel_l2(b.accum.cpu().numpy(), ob.accum) <= (1e-3 if scene == "cloud_vol" else 2e-6)      # value-only sky code in the vol_integrator


@pytest.mark.parametrize("scene", ["dragon sun", "dragon sun+sky", "sphere close by", "sphere on the way out", "point lights, deeper loops", "rotated volume"])
def test_convex_exit_changes_nothing(pkg, scene):
    """CONVEX EXIT (round 6, vpt_walk.h TRX; single-volume scenes, TIMED instantiations of the direct tracer): a walk that has taken a tracking step and stands in an
    empty node has left the box of non-empty leaves for good.  A ratio-tracking walk ends there (nothing ahead can change its value); a delta-tracking walk ends the PATH
    there when its ray's line misses the reference sphere by a wide margin (get_closest_object at :1806 would answer "nothing" from wherever the pushes end).  The COUNTING
    instantiations push on as the reference does (their skip counts are the oracle's): timed vs counting is therefore the A/B -- every buffer bit-identical, on views
    where the sphere is far, where it sits right next to the volume (many exits do NOT clear it: those push on), with point lights (eleven Tr walks per scatter), with
    volume_depth 2 / a short ray_depth, and with a rotated volume (its bounds are the box of its oriented box: still a box of leaves)."""
    import ctypes as C
    from vpt_amd.abi import Float3
    lib = pkg.load_library()
    if scene == "dragon sun":
        sd = pkg.scene.dragon_scene(192, 108, "sun")
    elif scene == "dragon sun+sky":
        sd = pkg.scene.dragon_scene(192, 108, "c2")
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    elif scene in ("sphere close by", "sphere on the way out"):
        sd = pkg.scene.dragon_scene(192, 108, "sun")
        lo, hi = Float3(), Float3()
        lib.vpt_gpu_vdb_bounds(C.byref(sd.volumes[0][0]), C.byref(lo), C.byref(hi))
        if scene == "sphere close by":
            sd.sphere.center = Float3(hi.x + 0.9, (lo.y + hi.y) * 0.5, (lo.z + hi.z) * 0.5)     # touching the padded root box: rays leave the volume straight at it
            sd.sphere.radius = 1.2
        else:
            o = sd.camera.origin
            sd.sphere.center = Float3(hi.x * 2.0 - o.x * 0.2, hi.y * 1.5, hi.z * 2.0 - o.z * 0.2)
            sd.sphere.radius = 2.5
    elif scene == "point lights, deeper loops":
        sd = pkg.scene.dragon_scene(160, 90, "c1")
        sd.kp.sun_mult = 1.0
        sd.kp.ray_depth = 4
        sd.kp.volume_depth = 2
    else:
        sd = pkg.scene.dragon_scene(192, 108, "sun")
        import numpy as np_
        vdb = sd.volumes[0][0]
        ang = 0.6
        rot = np_.array([[np_.cos(ang), 0, np_.sin(ang), 0], [0, 1, 0, 0], [-np_.sin(ang), 0, np_.cos(ang), 0], [0, 0, 0, 1]], np_.float32)
        m = np_.array([[vdb.xform[r][c] for c in range(4)] for r in range(4)], np_.float32) @ rot
        for r in range(4):
            for c in range(4):
                vdb.xform[r][c] = float(m[r, c])

    def run(counting):
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.ctx.set_counting(counting)
        hb.render(6)
        hb.sync()
        out = {b: getattr(hb, b).cpu().numpy().copy() for b in ("accum", "cost", "depth", "raw", "display", "blue_noise")}
        st = hb.ctx.stats()
        hb.ctx.close()
        return out, st
    a, sa = run(False)
    b, sb = run(True)
    assert np.isfinite(a["accum"]).all() and a["accum"].max() > 0 and sb.skip_steps > 0 and sb.tracking_steps > 0
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg="%s: %s" % (scene, k))
    assert sa.queued_rays == sb.queued_rays


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_convex_exit_on_random_views(pkg, seed):
    """test_convex_exit_changes_nothing on SEEDED RANDOM set-ups instead of hand-picked ones: camera anywhere around (and sometimes inside) the dragon's box, the
    reference sphere anywhere from touching the box to far away -- and small enough, half of the time, that exits graze it: the robust "clears the sphere" test and the
    `B == 0` rule decide --, the sun anywhere above the horizon, ray_depth 1-4, volume_depth 1-3, thin and dense media, the volume rotated about y in a third of the cases.
    Timed (exits taken) against counting (every push walked, as the reference does): every buffer bit-identical, the same rays queued."""
    import ctypes as C
    from vpt_amd.abi import Float3
    lib = pkg.load_library()
    rs = np.random.RandomState(100 + seed)
    exits_possible = 0
    for case in range(8):
        w, h = int(rs.choice([192, 160, 131])), int(rs.choice([108, 90, 77]))
        sd = pkg.scene.dragon_scene(w, h, "c1" if case % 4 == 3 else "sun")
        if case % 4 == 3:
            sd.kp.sun_mult = 1.0                            # point lights AND the sun: twelve Tr walks per scatter
        vdb = sd.volumes[0][0]
        if case % 3 == 1:
            ang = float(rs.uniform(-1.2, 1.2))
            rot = np.array([[np.cos(ang), 0, np.sin(ang), 0], [0, 1, 0, 0], [-np.sin(ang), 0, np.cos(ang), 0], [0, 0, 0, 1]], np.float32)
            m = np.array([[vdb.xform[r][c] for c in range(4)] for r in range(4)], np.float32) @ rot
            for r in range(4):
                for c in range(4):
                    vdb.xform[r][c] = float(m[r, c])
        lo, hi = Float3(), Float3()
        lib.vpt_gpu_vdb_bounds(C.byref(vdb), C.byref(lo), C.byref(hi))
        ctr = np.array([(lo.x + hi.x) * 0.5, (lo.y + hi.y) * 0.5, (lo.z + hi.z) * 0.5])
        half = np.array([hi.x - lo.x, hi.y - lo.y, hi.z - lo.z]) * 0.5
        size = float(np.linalg.norm(half))
        d = rs.normal(size=3); d /= np.linalg.norm(d); d[1] = abs(d[1]) * 0.7
        eye = ctr + d * size * float(rs.uniform(0.2 if case == 5 else 1.3, 5.0))
        look = ctr + rs.uniform(-0.6, 0.6, 3) * half
        lib.vpt_camera_update(C.byref(sd.camera), Float3(*[float(v) for v in eye]), Float3(*[float(v) for v in look]), Float3(0, 1, 0), float(rs.uniform(20.0, 70.0)), w / h, 0.0)
        sdir = rs.normal(size=3); sdir /= np.linalg.norm(sdir)
        sd.sphere.radius = float(rs.uniform(0.15, 0.6) if case % 2 else rs.uniform(0.8, 3.0))
        sdist = float(np.max(half)) + sd.sphere.radius * float(rs.uniform(0.8, 1.3) if case % 2 == 0 else rs.uniform(1.0, 4.0))     # from overlapping the box's edge to well away
        sc = ctr + sdir * sdist
        sd.sphere.center = Float3(float(sc[0]), float(sc[1]), float(sc[2]))
        sd.kp.azimuth = float(rs.uniform(0.0, 360.0))
        sd.kp.elevation = float(rs.uniform(2.0, 88.0))
        sd.kp.ray_depth = int(rs.randint(1, 5))
        sd.kp.volume_depth = int(rs.randint(1, 4))
        sd.kp.density_mult = float(sd.kp.density_mult) * float(rs.choice([0.25, 1.0, 3.0]))

        def run(counting):
            hb = pkg.scene.HipBinding(sd, device=0)
            hb.ctx.set_counting(counting)
            hb.render(4)
            hb.sync()
            out = {b: getattr(hb, b).cpu().numpy().copy() for b in ("accum", "cost", "depth", "raw", "display", "blue_noise")}
            st = hb.ctx.stats()
            hb.ctx.close()
            return out, st
        a, sa = run(False)
        b, sb = run(True)
        assert np.isfinite(a["accum"]).all()
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg="seed %d case %d: %s" % (seed, case, k))
        assert sa.queued_rays == sb.queued_rays
        exits_possible += sb.skip_steps > 0 and sb.tracking_steps > 0
        print("seed %d case %d: %d rays, %d steps, %d skips" % (seed, case, sb.queued_rays, sb.tracking_steps, sb.skip_steps))
    assert exits_possible >= 6, exits_possible


def _skip_if_stale(lib):
    """a study library left over from an earlier state of the sources (it is git-ignored and built by hand) may lack entry points the Python host binds"""
    import ctypes
    import __graft_entry__ as ge
    try:
        h = ctypes.CDLL(lib)
    except OSError as e:
        pytest.skip("study library %s does not load: %s" % (os.path.basename(lib), e))
    missing = [sym for sym in ge.load_package().ABI_SYMBOLS if not hasattr(h, sym)]
    if missing:
        pytest.skip("stale study library %s (lacks %s): rebuild it" % (os.path.basename(lib), ", ".join(missing[:3])))


def test_zero_footprint_mask_is_bit_identical():
    """The zero-footprint mask (round 6, DVolume::zmask: one bit per block of footprint origins, set when every footprint of the block is eight exact zeros -- such a
    look-up is +0 whatever its weights, so the two quad loads are skipped) is exact and measured SLOWER than the loads it saves on every config (its own dependent load
    ahead of the quads: profiles/r06_zero_mask.txt), so it is NOT part of the product library.  Where its study library exists
    (`python volumetric-path-tracer_amd/build.py --variant zmask -DVPT_ZERO_MASK`), tools/zmask_ab.py renders four scenes with block edges 4 and 8 (odd extents: partial edge
    blocks, clamped footprints on every face): every buffer and every count equal to the render without a mask, the mask really answers look-ups, the image is the oracle's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "volumetric-path-tracer_amd", "libvpt_hip_zmask.so")
    if not os.path.exists(lib):
        pytest.skip("study library libvpt_hip_zmask.so not built (build.py --variant zmask -DVPT_ZERO_MASK)")
    _skip_if_stale(lib)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "zmask_ab.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, VPT_LIB_PATH=lib), cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "8 cases bit-identical" in r.stdout


@pytest.mark.parametrize("layout", [None, "bricks", "quads"])
def test_24_bit_index_arithmetic_is_bit_identical(pkg, monkeypatch, layout):
    """texel indices are formed with the 24-bit multiplier where the grid extents allow it (vpt_trace_common.h imul);
    the 32-bit path (VPT_NO_ADDR24, what a >16.7 M-row grid would take) must give the same bits in every grid layout"""
    if layout:
        monkeypatch.setenv("VPT_RELAID_MIN_BYTES", "0")
        monkeypatch.setenv("VPT_GRID_LAYOUT", layout)
    sd = pkg.scene.instanced_scene(96, 64, n=18, grid=3, aperture=0.3)
    a = pkg.scene.HipBinding(sd, device=0)
    a.render(3); a.sync()
    monkeypatch.setenv("VPT_NO_ADDR24", "1")
    b = pkg.scene.HipBinding(sd, device=0)
    b.render(3); b.sync()
    assert a.accum.abs().max() > 0
    np.testing.assert_array_equal(a.accum.cpu().numpy(), b.accum.cpu().numpy())
    np.testing.assert_array_equal(a.depth.cpu().numpy(), b.depth.cpu().numpy())


@pytest.mark.parametrize("aperture", [0.0, 2.0])
def test_sample_heads_are_bit_identical_to_full_records(pkg, monkeypatch, aperture):
    """untraced samples travel as 16-byte heads (+ a 16-byte origin when the lens is open) instead of 64-byte records
    (record stream); VPT_NO_HEADS restores the full records: same bits either way"""
    sd = pkg.scene.instanced_scene(96, 64, n=18, grid=3, aperture=aperture)
    a = pkg.scene.HipBinding(sd, device=0)
    a.render(3); a.sync()
    monkeypatch.setenv("VPT_NO_HEADS", "1")
    b = pkg.scene.HipBinding(sd, device=0)
    b.render(3); b.sync()
    assert a.accum.abs().max() > 0
    for buf in ("accum", "depth", "raw", "display"):
        np.testing.assert_array_equal(getattr(a, buf).cpu().numpy(), getattr(b, buf).cpu().numpy())


def test_striped_batches_spanning_chunks(pkg, monkeypatch):
    """iteration striping (stride 3, starting at iteration 1) across several record chunks == one chunk,
    and == the oracle rendering the same stripe"""
    import oracle_binding
    sd = pkg.scene.dragon_scene(72, 40, "sun")
    phi = np.float32((1.0 + np.sqrt(np.float32(5.0))) / np.float32(2.0))
    monkeypatch.setenv("VPT_BATCH_ITERS", "2")
    a = pkg.scene.HipBinding(sd, device=0)
    a.ctx.blue_noise_advance(a.blue_noise, 1, sd.width * sd.height)
    a.render(5, iter_stride=3, iteration=1)
    a.sync()
    monkeypatch.delenv("VPT_BATCH_ITERS")
    b = pkg.scene.HipBinding(sd, device=0)
    b.ctx.blue_noise_advance(b.blue_noise, 1, sd.width * sd.height)
    b.render(5, iter_stride=3, iteration=1)
    b.sync()
    np.testing.assert_array_equal(a.accum.cpu().numpy(), b.accum.cpu().numpy())
    np.testing.assert_array_equal(a.blue_noise.cpu().numpy(), b.blue_noise.cpu().numpy())
    ob = oracle_binding.OracleBinding(sd)
    live = min(sd.width * sd.height, 65536)       # a launch advances only the entries its pixels own
    ob.blue_noise[:live, :] = np.fmod(ob.blue_noise[:live] + phi, np.float32(1.0))
    ob.render(5, iter_stride=3, iteration=1)
    assert rel_l2(a.accum.cpu().numpy(), ob.accum) <= 2e-6
    np.testing.assert_array_equal(a.blue_noise.cpu().numpy(), ob.blue_noise)


def test_pool_tracer_is_bit_identical_to_lane_tracer():
    """The round-3 pool tracer (csrc/variants/vpt_trace_pool.hip: direct_integrator with the rays in an LDS pool per CU, waves claim
    phase-homogeneous batches; measured slower than the lane-bound tracer) is NOT part of the product library.  Where its study
    library exists (`python volumetric-path-tracer_amd/build.py --variant pool --with-pool`), tools/pool_ab.py renders four scenes with both
    tracers of that library: every buffer and every look-up / step / skip count bit-identical."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "volumetric-path-tracer_amd", "libvpt_hip_pool.so")
    if not os.path.exists(lib):
        pytest.skip("study library libvpt_hip_pool.so not built (build.py --variant pool --with-pool)")
    _skip_if_stale(lib)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pool_ab.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, VPT_LIB_PATH=lib), cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "4 scenes bit-identical" in r.stdout


@pytest.mark.parametrize("scene", ["dragon", "fireball", "instanced", "cloud_vol"])
def test_quotient_by_checked_reciprocal_is_the_division(pkg, monkeypatch, scene):
    """to_unit divides the index-space position by the grid extent (render_kernel.cu:996).  Where the host has checked the
    extent over a whole binade (csrc/vpt_fastdiv.h) the look-ups multiply by its rounded reciprocal and correct with one exact
    residual instead; VPT_NO_FAST_DIV keeps the division.  Same bits: every buffer and every count must be identical."""
    def make():
        if scene == "dragon":
            return pkg.scene.dragon_scene(160, 90, "sun")
        if scene == "fireball":
            return pkg.scene.fireball_scene(96, 64, n=37)                 # emission grid: a second extent per look-up
        if scene == "instanced":
            return pkg.scene.instanced_scene(96, 64, n=18, grid=3, aperture=0.3)    # instance loop, colour grids
        sd = pkg.scene.cloud_scene(96, 64, shape=(76, 44, 64), env=(128, 64))        # vol_integrator, HDRI
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
        return sd
    sd = make()
    a = pkg.scene.HipBinding(sd, device=0)
    a.ctx.set_counting(True)
    a.render(5); a.sync()
    sa = a.ctx.stats()
    monkeypatch.setenv("VPT_NO_FAST_DIV", "1")
    b = pkg.scene.HipBinding(sd, device=0)
    b.ctx.set_counting(True)
    b.render(5); b.sync()
    sb = b.ctx.stats()
    assert a.accum.abs().max() > 0
    for buf in ("accum", "depth", "raw", "display"):
        np.testing.assert_array_equal(getattr(a, buf).cpu().numpy(), getattr(b, buf).cpu().numpy())
    for k in ("samples", "density_lookups", "color_lookups", "emission_lookups", "tracking_steps", "skip_steps", "queued_rays"):
        assert getattr(sa, k) == getattr(sb, k), k


@pytest.mark.parametrize("size", [(160, 90), (97, 61)])
def test_image_plane_quotients_by_checked_reciprocal(pkg, monkeypatch, size):
    """volume_rt_kernel's u = (x + jitter) / width, v = (y + jitter) / height (render_kernel.cu:2243-2244) are formed like the look-up's
    quotient (csrc/vpt_trace.hip raygen; extents checked by the host, a zero numerator divides): same rays, bit for bit -- closed and open lens."""
    for config, aperture in (("sun", 0.0), ("sun", 0.4)):
        sd = pkg.scene.dragon_scene(size[0], size[1], config)
        if aperture:
            sd.camera.lens_radius = aperture / 2
        monkeypatch.delenv("VPT_NO_FAST_DIV", raising=False)
        a = pkg.scene.HipBinding(sd, device=0)
        a.render(4); a.sync()
        monkeypatch.setenv("VPT_NO_FAST_DIV", "1")
        b = pkg.scene.HipBinding(sd, device=0)
        b.render(4); b.sync()
        assert a.accum.abs().max() > 0
        for buf in ("accum", "depth", "raw", "display", "blue_noise"):
            np.testing.assert_array_equal(getattr(a, buf).cpu().numpy(), getattr(b, buf).cpu().numpy())


# ---- round 5: host-side scheduling that must change no bit -----------------------------------------------------------------------------

def _frame_scene(pkg, kind):
    import ctypes as C
    if kind == "dragon sun+sky":
        sd = pkg.scene.dragon_scene(256, 144, "c2")
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    elif kind == "dragon sun only":
        sd = pkg.scene.dragon_scene(256, 144, "sun")
    elif kind == "fireball sun+sky":
        sd = pkg.scene.fireball_scene(192, 108, n=48, sky=True)
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    elif kind == "instances open lens":
        sd = pkg.scene.instanced_scene(256, 144, n=32, grid=3, aperture=2.0, sky=True)
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    else:
        sd = pkg.scene.cloud_scene(192, 108, shape=(48, 40, 56), env=(128, 64), integrator=1)
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    return sd


@pytest.mark.parametrize("kind", ["dragon sun+sky", "dragon sun only", "fireball sun+sky", "instances open lens", "cloud vol_integrator", "dragon sun+sky, render off at 12"])
def test_frame_ahead_changes_nothing(pkg, monkeypatch, kind):
    """FRAME-AHEAD (csrc/vpt_ctx.h): from the second identical one-iteration call on, vpt_render traces the rays of the next 2, 4, 8, 16 iterations in
    one raygen + tracer launch and the following calls run only their tail.  Against VPT_NO_FRAME_AHEAD=1 (one launch per frame, as rounds 1-4):
    EVERY buffer after EVERY frame is bit-identical -- accumulation, cost, depth, raw, display and the caller's blue-noise state -- through the growth
    of the batches, a camera move in the middle of a batch (what was traced ahead is discarded), and a return to the first camera."""
    import ctypes as C
    from vpt_amd.abi import Float3
    sd = _frame_scene(pkg, kind.split(",")[0])
    if "render off" in kind:
        sd.kp.max_interactions = 12                  # (iterations from 12 on are "not rendered": WHITE samples, :2248 -- the boundary falls inside a batch traced ahead)
    lib = pkg.load_library()
    frames, move_at, back_at = 23, 9, 14

    def run():
        hb = pkg.scene.HipBinding(sd, device=0)
        cam0 = type(hb.sd.camera).from_buffer_copy(hb.sd.camera)
        out = []
        for f in range(frames):
            if f == move_at:
                o = hb.sd.camera.origin
                lib.vpt_camera_update(C.byref(hb.sd.camera), Float3(o.x * 0.9, o.y * 1.05, o.z * 0.95), Float3(0.0, 0.0, 0.0), Float3(0, 1, 0), 40.0,
                                      sd.width / sd.height, float(2.0 * hb.sd.camera.lens_radius))
            if f == back_at:
                C.memmove(C.byref(hb.sd.camera), C.byref(cam0), C.sizeof(cam0))
            hb.render_frame()
            hb.sync()
            out.append({b: getattr(hb, b).cpu().numpy().copy() for b in ("accum", "cost", "depth", "raw", "display", "blue_noise")})
        C.memmove(C.byref(hb.sd.camera), C.byref(cam0), C.sizeof(cam0))
        hb.ctx.close()
        return out
    a = run()
    monkeypatch.setenv("VPT_NO_FRAME_AHEAD", "1")
    b = run()
    assert np.isfinite(a[-1]["accum"]).all() and a[-1]["accum"].max() > 0
    for f in range(frames):
        for k in a[f]:
            np.testing.assert_array_equal(a[f][k], b[f][k], err_msg="frame %d, %s" % (f, k))


def test_frame_ahead_invalidate_after_an_in_place_edit(pkg, monkeypatch):
    """include/vpt_abi.h "FRAME-AHEAD": device memory rewritten IN PLACE behind an unchanged pointer is invisible to the key of the rays traced ahead.  A host that
    rewrites its emission table between two frames calls vpt_frame_ahead_invalidate: the sequence then equals frame by frame bit for bit (without the call, frames
    already traced would run their tails on samples that saw the old table); vpt_set_frame_ahead(ctx, 0) does the same for good."""
    sd = _frame_scene(pkg, "fireball sun+sky")
    frames, edit_at = 14, 6

    def run(mode):
        hb = pkg.scene.HipBinding(sd, device=0)
        if mode == "off":
            hb.ctx.set_frame_ahead(False)
        out = []
        for f in range(frames):
            if f == edit_at:
                hb.sync()
                hb.emission_lut.mul_(0.25)                       # in place: same pointer in kernel_params
                import torch
                torch.cuda.synchronize()
                if mode == "invalidate":
                    hb.ctx.frame_ahead_invalidate()
            hb.render_frame()
            hb.sync()
            out.append({b: getattr(hb, b).cpu().numpy().copy() for b in ("accum", "depth", "raw", "display", "blue_noise")})
        hb.ctx.close()
        return out
    a = run("invalidate")
    c = run("off")
    monkeypatch.setenv("VPT_NO_FRAME_AHEAD", "1")
    b = run("plain")
    assert np.isfinite(a[-1]["accum"]).all() and a[-1]["accum"].max() > 0
    assert not np.array_equal(a[edit_at - 1]["accum"], a[-1]["accum"])
    for f in range(frames):
        for k in a[f]:
            np.testing.assert_array_equal(a[f][k], b[f][k], err_msg="invalidate: frame %d, %s" % (f, k))
            np.testing.assert_array_equal(c[f][k], b[f][k], err_msg="set_frame_ahead(0): frame %d, %s" % (f, k))


@pytest.mark.parametrize("kind", ["instances open lens", "dragon sun+sky"])
def test_frame_ahead_after_a_batch(pkg, monkeypatch, kind):
    """A batch render followed by the per-frame call (advisor, round 5: the combination no test covered): render(7) builds the per-view caches -- behind the open
    lens the lens domes, so raygen resolves the untraced samples itself (LENSRES) and the tracer resolves the finished paths; behind the closed one the compact
    32-byte ray records -- and the 12 one-iteration calls after it are served from rays traced ahead over those caches: one raygen + tracer launch per 2, 4, 8
    iterations, sky_fix over the whole ahead batch, a tail per slice with its td / heads / origin offsets.  Every buffer after the batch and after every frame
    equals VPT_NO_FRAME_AHEAD=1, and (open lens) VPT_NO_LENS_LEAN=1."""
    sd = _frame_scene(pkg, kind)

    def run():
        hb = pkg.scene.HipBinding(sd, device=0)
        out = []
        hb.render(7)
        hb.sync()
        out.append({b: getattr(hb, b).cpu().numpy().copy() for b in ("accum", "cost", "depth", "raw", "display", "blue_noise")})
        for _ in range(12):
            hb.render_frame()
            hb.sync()
            out.append({b: getattr(hb, b).cpu().numpy().copy() for b in ("accum", "cost", "depth", "raw", "display", "blue_noise")})
        hb.ctx.close()
        return out
    monkeypatch.delenv("VPT_BATCH_ITERS", raising=False)
    a = run()
    assert np.isfinite(a[-1]["accum"]).all() and a[-1]["accum"].max() > 0
    variants = ["VPT_NO_FRAME_AHEAD"] + (["VPT_NO_LENS_LEAN"] if "open lens" in kind else ["VPT_NO_COMPACT_RAYS"])
    for sw in variants:
        monkeypatch.setenv(sw, "1")
        b = run()
        monkeypatch.delenv(sw)
        for f in range(len(a)):
            for k in a[f]:
                np.testing.assert_array_equal(a[f][k], b[f][k], err_msg="%s: step %d, %s" % (sw, f, k))


@pytest.mark.parametrize("kind", ["dragon sun+sky", "dragon sun only", "fireball sun+sky", "cloud vol_integrator"])
def test_compact_ray_records_change_nothing(pkg, monkeypatch, kind):
    """Behind a closed lens a queued ray's record is 32 bytes -- {position reached | (t_hit, depth, t_box), packed word} + the Philox block -- instead of 64: the
    origin is the camera's, the direction is the sample's head, the Philox counter follows from the iteration (csrc/vpt_device.h: TraceParams::compact_rays;
    vpt_trace_common.h: load_ray_record rebuilds the 64-byte layout's values).  Against VPT_NO_COMPACT_RAYS=1: every buffer and every count identical, counting
    build (it carries raygen's push count in the word) and timed build, chunked batches, frames."""
    sd = _frame_scene(pkg, kind)
    monkeypatch.setenv("VPT_BATCH_ITERS", "3")

    def run(counting):
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.ctx.set_counting(counting)
        hb.render(7)
        hb.render_frame()
        hb.render(2)
        hb.sync()
        st = hb.ctx.stats()
        out = {b: getattr(hb, b).cpu().numpy().copy() for b in ("accum", "cost", "depth", "raw", "display", "blue_noise")}
        hb.ctx.close()
        return out, st
    res = {}
    for counting in (True, False):
        res[counting] = run(counting)
    monkeypatch.setenv("VPT_NO_COMPACT_RAYS", "1")
    for counting in (True, False):
        a, sa = res[counting]
        b, sb = run(counting)
        assert np.isfinite(a["accum"]).all() and a["accum"].max() > 0
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg="%s (counting %s)" % (k, counting))
        if counting:
            for c in ("samples", "density_lookups", "color_lookups", "emission_lookups", "tracking_steps", "skip_steps", "queued_rays"):
                assert getattr(sa, c) == getattr(sb, c), c


@pytest.mark.parametrize("kind", ["dragon sun+sky", "dragon sun only", "fireball sun+sky", "instances closed lens"])
def test_queue_of_pieces_changes_nothing(pkg, monkeypatch, kind):
    """With the compact records in queue order the direct tracer's queue holds PIECES -- {first record, count} of consecutive records, cut by raygen into sizes that shrink
    towards the end of the launch -- instead of one entry per ray (csrc/vpt_device.h: TraceParams::piece_max; a claim is one piece).  Which wave traces which ray is all
    that moves.  Against VPT_PIECE_MAX=0 (a queue of entries) and against other piece sizes: every buffer and every count identical, counting and timed builds, chunked
    batches, frames, a launch of many iterations (pieces of several sizes)."""
    if kind == "instances closed lens":
        sd = pkg.scene.instanced_scene(256, 144, n=32, grid=3, aperture=0.0, sky=True)
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    else:
        sd = _frame_scene(pkg, kind)

    def run(counting, batch_iters):
        if batch_iters:
            monkeypatch.setenv("VPT_BATCH_ITERS", str(batch_iters))
        else:
            monkeypatch.delenv("VPT_BATCH_ITERS", raising=False)
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.ctx.set_counting(counting)
        hb.render(7 if batch_iters else 40)
        hb.render_frame()
        hb.render(2)
        hb.sync()
        st = hb.ctx.stats()
        out = {b: getattr(hb, b).cpu().numpy().copy() for b in ("accum", "cost", "depth", "raw", "display", "blue_noise")}
        hb.ctx.close()
        return out, st
    cases = [(True, 3), (False, 3), (False, 0)]
    res = {c: run(*c) for c in cases}
    assert res[cases[0]][1].queued_rays > 0
    for sw, val in (("VPT_PIECE_MAX", "0"), ("VPT_PIECE_MAX", "64"), ("VPT_PIECE_DIV", "1")):
        monkeypatch.setenv(sw, val)
        for c in cases:
            a, sa = res[c]
            b, sb = run(*c)
            assert np.isfinite(a["accum"]).all() and a["accum"].max() > 0
            for k in a:
                np.testing.assert_array_equal(a[k], b[k], err_msg="%s=%s: %s %s" % (sw, val, k, c))
            for cn in ("samples", "queued_rays") + (("density_lookups", "color_lookups", "emission_lookups", "tracking_steps", "skip_steps") if c[0] else ()):
                assert getattr(sa, cn) == getattr(sb, cn), (sw, val, cn)
        monkeypatch.delenv(sw)


@pytest.mark.parametrize("kind", ["instances open lens", "dragon open lens", "dragon open lens, sphere in view", "dragon open lens, render off at 3"])
def test_resolved_samples_behind_an_open_lens_change_nothing(pkg, monkeypatch, kind):
    """RESOLVED SAMPLES behind an OPEN lens (round 5): every sample starts somewhere on the lens disc, and the dome that serves its environment term is the one of
    its origin's table variant (csrc/vpt_dome.h: dome_variant).  The tracer resolves finished paths from it, raygen resolves the UNTRACED samples from it (their head
    then carries the value: no origin stream), sky_fix_kernel evaluates in full what no dome serves, and the tail only streams.  Against VPT_NO_LENS_LEAN=1 -- 64-byte
    records, heads + origins, the environment added inside the tail's per-pixel loop: the same operations on the same values, so every buffer and count is identical,
    counting and timed builds, chunked batches, frames (VPT_BATCH_ITERS switches frame-ahead off here: the two together are test_frame_ahead_after_a_batch)."""
    import ctypes as C
    from vpt_amd.abi import Float3
    lib = pkg.load_library()
    if kind.startswith("instances"):
        sd = _frame_scene(pkg, "instances open lens")
    else:
        sd = pkg.scene.dragon_scene(256, 144, "c2")
        o = sd.camera.origin
        lib.vpt_camera_update(C.byref(sd.camera), Float3(o.x, o.y, o.z), Float3(0.0, 0.0, 0.0), Float3(0, 1, 0), 30.0, 256 / 144, 1.5)
        if "sphere" in kind:
            sd.sphere.center = Float3(o.x * 0.55, o.y * 0.55 + 1.0, o.z * 0.55 - 2.0)
            sd.sphere.radius = 1.5
        if "render off" in kind:
            sd.kp.max_interactions = 3
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    monkeypatch.setenv("VPT_BATCH_ITERS", "3")
    lib.vpt_test_get_cache_state.argtypes = [C.c_void_p, C.POINTER(C.c_int)]

    def run(counting):
        hb = pkg.scene.HipBinding(sd, device=0)
        hb.ctx.set_counting(counting)
        hb.render(7)
        state = (C.c_int * 8)()
        assert lib.vpt_test_get_cache_state(hb.ctx.h, state) == 0
        for _ in range(4):
            hb.render_frame()
        hb.render(2)
        hb.sync()
        st = hb.ctx.stats()
        out = {b: getattr(hb, b).cpu().numpy().copy() for b in ("accum", "cost", "depth", "raw", "display", "blue_noise")}
        hb.ctx.close()
        return out, st, list(state)
    res = {c: run(c) for c in (True, False)}
    assert res[False][2][2] and res[False][2][6], res[False][2]           # lens domes + resolved samples in use
    monkeypatch.setenv("VPT_NO_LENS_LEAN", "1")
    for counting in (True, False):
        a, sa, _ = res[counting]
        b, sb, cb = run(counting)
        assert cb[2] and not cb[6], cb
        assert np.isfinite(a["accum"]).all() and a["accum"].max() > 0
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg="%s (counting %s)" % (k, counting))
        if counting:
            for c in ("samples", "density_lookups", "color_lookups", "tracking_steps", "skip_steps", "queued_rays"):
                assert getattr(sa, c) == getattr(sb, c), c
