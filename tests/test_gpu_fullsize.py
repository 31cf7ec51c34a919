"""Parity at the BASELINE sizes (round-1 verdict, "Parity at the BASELINE sizes"): ONE iteration of every BASELINE config at
its specified resolution, grid sizes and instance count (config 4 at half the linear grid size: the 3.5 GB grid is a
GPU-side generator; 0.44 GB keeps the host copy and the CPU walk within seconds), compared over the WHOLE frame with the
oracle running on the GPU box's host cores:

  * depth and alpha buffers bit-identical (they are functions of the walk decisions alone),
  * the exact reference-defined look-up / step / skip counts of the frame,
  * accum within the 1e-3 relative-L2 tolerance (the sky / environment value is value-only arithmetic),

plus, for config 2, several consecutive iterations through the per-frame entry point vpt_render.  What the small scenes of
the other test files cannot show is covered here: 2 M / 8.3 M pixel record streams, the 24-bit index path on 128^3 - 608^3
grids, corner-quad density (config 4 at this size is above VPT_RELAID_MIN_BYTES), 100 instances over the octree with per-sub-cell
candidate lists, the thin lens at 4K, the real atmosphere tables.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(1e-30, np.sqrt((b ** 2).sum())))


def _per_pixel(got, ref):
    """relative error per pixel (largest channel difference over the pixel's brightest channel), over the pixels above 1e-3"""
    a = np.asarray(got, np.float64); b = np.asarray(ref, np.float64)
    lum = b.max(1)
    m = lum > 1e-3
    return np.abs(a - b).max(1)[m] / lum[m]


def _cache_state(pkg, hb):
    """which per-view caches of the environment tail the last render used (include/vpt_testhooks.h)"""
    import ctypes as C
    lib = pkg.load_library()
    out = (C.c_int * 8)()
    lib.vpt_test_get_cache_state.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    assert lib.vpt_test_get_cache_state(hb.ctx.h, out) == 0
    return dict(zip(("sky_patch", "never_traced", "sky_dome", "dome_variants", "cam_table", "dir_table", "resolved_samples", "leaf_tiles"), list(out)[:8]))


def _compare(pkg, sd, iterations=1, tol=1e-3, per_frame=False, p99=2e-3, worst=1e-2, caches=None, median=None):
    """caches: names of the per-view caches (see _cache_state) the render MUST have used -- a batch of >= 2 iterations is what
    bench.py times, and it goes through the sky patches, the never-traced pixel mask and the sky dome(s); one iteration does not."""
    import oracle_binding
    hb = pkg.scene.HipBinding(sd, device=0)
    ob = oracle_binding.OracleBinding(sd)
    hb.ctx.set_counting(True)
    if per_frame:
        for _ in range(iterations):
            hb.render_frame()
    else:
        hb.render(iterations)
    hb.sync()
    st = hb.ctx.stats()
    if caches is not None:
        state = _cache_state(pkg, hb)
        for c in caches:
            assert state[c], (c, state)
    ob.render(iterations)
    got = hb.accum.cpu().numpy()
    assert np.isfinite(got).all() and ob.accum.max() > 0
    np.testing.assert_array_equal(hb.depth.cpu().numpy(), ob.depth)                       # bit for bit
    np.testing.assert_array_equal(hb.raw.cpu().numpy()[:, 3], ob.raw[:, 3])               # alpha of the last iteration, bit for bit
    e = rel_l2(got, ob.accum)
    assert e <= tol, e
    # ... and per PIXEL, not only per image: the value-only sky code (approximate divide / root, fp32 weights, the per-frame tables)
    # against the oracle's strict arithmetic.  Measured on one iteration (profiles/r03_sky_error_probe.txt): config 2 median 7e-6,
    # 99th percentile 6e-4, worst pixel 7e-3 (the same with the ground table switched off: the outliers are ground points whose
    # binary32 radius flips between the two arithmetics, not the table); config 3 6e-4 / 2e-3; config 5 (open lens: table variants)
    # 3.7e-3 / 5.2e-3 after ONE iteration -- 7e-4 / 5.2e-3 without its ground tables; after the two iterations the tests render 1.98e-3 / 4.7e-3.  (Over a few iterations the 99th percentile first RISES -- 1.5e-3
    # after three frames of config 2: more pixels have met one outlier sample, each diluted by the mean -- before it falls.)
    rel = _per_pixel(got, ob.accum)
    assert rel.size > 0.5 * got.shape[0] or ob.accum.max(1).mean() < 1e-3
    if rel.size:
        assert np.quantile(rel, 0.99) <= p99, np.quantile(rel, [0.5, 0.99, 0.999])
        assert rel.max() <= worst, rel.max()
        if median is not None:
            assert np.quantile(rel, 0.5) <= median, np.quantile(rel, [0.5, 0.95, 0.99])
    if not per_frame:                                # a per-frame sequence reports the counts of its last launch only
        assert st.samples == ob.stats.samples == sd.width * sd.height * iterations
        for c in ("density_lookups", "color_lookups", "emission_lookups", "tracking_steps", "skip_steps"):
            assert getattr(st, c) == getattr(ob.stats, c), (c, getattr(st, c), getattr(ob.stats, c))
    disp = hb.display.cpu().numpy().view(np.uint32)
    diff = np.abs(((disp[:, None] >> np.array([16, 8, 0])) & 255).astype(int) - ((ob.display[:, None] >> np.array([16, 8, 0])) & 255).astype(int))
    assert diff.max() <= 2
    hb.ctx.close()
    return e, st


def test_config2_dragon_1080p(pkg):
    sd = pkg.scene.dragon_scene(1920, 1080, "c2")
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    e, st = _compare(pkg, sd, 1)
    assert st.density_lookups > 0 and st.skip_steps > 0
    # what bench.py times: a BATCH, which builds and uses the per-view caches (patches for the untraced samples, the mask of
    # never-traced pixels that raygen skips, the dome for the traced ones) -- whole frame against the oracle, depth / alpha / counts exact
    # (resolved_samples: the tracer adds a finished path's environment term itself and the tail only streams -- round 4's largest change to the
    # tail; required here so that a silent fall-back to tail_resolve_kernel cannot pass the whole-frame comparison)
    _compare(pkg, sd, 2, caches=("sky_patch", "never_traced", "sky_dome", "cam_table", "dir_table", "resolved_samples"))
    # and the literal drop-in call: three frames through vpt_render (the still view repeats: frames 2 and 3 use the caches too)
    _compare(pkg, sd, 3, per_frame=True, caches=("sky_patch", "never_traced", "sky_dome", "resolved_samples"))


def test_config3_fireball_1080p_sun_and_sky(pkg):
    sd = pkg.scene.fireball_scene(1920, 1080, n=256, sky=True)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    e, st = _compare(pkg, sd, 2, caches=("sky_patch", "sky_dome", "cam_table", "dir_table", "resolved_samples"))       # (its box fills the frame: no pixel is skipped)
    assert st.emission_lookups > 0


def _cloud_scene_1080p(pkg, shape):
    """config 4's stand-in with a HOST copy of the (GPU-generated) grid, so that HIP and oracle render the same voxels"""
    import torch
    gdev = pkg.scene.cloud_grid_torch(shape, device=torch.device("cuda", 0))
    grid = gdev.cpu().numpy()                          # one grid for both sides
    del gdev
    torch.cuda.empty_cache()
    S = pkg.scene
    sd = S.SceneDesc()
    sd.width, sd.height = 1920, 1080
    voxel = 40.0 / shape[2]
    vdb = S.make_gpu_vdb(grid, (0, 0, 0), (shape[2] - 1, shape[1] - 1, shape[0] - 1), S._grid_matrix(grid.shape, voxel), voxel)
    sd.volumes.append((vdb, grid, None, None))
    lib = pkg.load_library()
    sd.camera, _, _ = S.frame_camera(lib, [vdb], 1920, 1080)
    kp = S._base_kp(lib, 1920, 1080)
    kp.environment_type = 1
    kp.integrator = 1
    sd.kp = kp
    sd.env_map = S.hdri_map(2048, 1024)
    S._finish(sd)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    return sd, grid


def test_config4_cloud_half_size_grid_1080p(pkg):
    sd, grid = _cloud_scene_1080p(pkg, (608, 352, 512))
    assert grid.nbytes >= (8 << 20)                   # above VPT_RELAID_MIN_BYTES: the corner-quad layout is what runs
    e, st = _compare(pkg, sd, 1)
    assert st.tracking_steps > st.density_lookups      # vol_integrator's runs of empty sample() calls


def test_config4_cloud_benchmark_size_grid_1080p(pkg, monkeypatch):
    """The grid bench.py's config 4 renders, 1024x704x1216 (3.5 GB): its corner quads are 14 GB, addressed with byte offsets
    far above 4 GiB (vpt_trace_common.h: fetch_f32_quads).  One iteration of the whole 1080p frame:
      * corner quads vs the dense layout (VPT_GRID_LAYOUT=dense, 32-bit texel indices): every buffer and count bit-identical,
      * against the oracle walking a host copy of the same grid: depth and alpha bit-identical, counts equal, accum <= 1e-3."""
    import oracle_binding
    sd, grid = _cloud_scene_1080p(pkg, (1216, 704, 1024))
    assert grid.nbytes * 4 > (4 << 30)
    a = pkg.scene.HipBinding(sd, device=0)
    a.ctx.set_counting(True)
    a.render(1); a.sync()
    sa = a.ctx.stats()
    bufs = {k: getattr(a, k).cpu().numpy().copy() for k in ("accum", "depth", "raw", "display")}
    a.ctx.close()
    del a
    monkeypatch.setenv("VPT_GRID_LAYOUT", "dense")
    b = pkg.scene.HipBinding(sd, device=0)
    b.ctx.set_counting(True)
    b.render(1); b.sync()
    sb = b.ctx.stats()
    for k, v in bufs.items():
        np.testing.assert_array_equal(v, getattr(b, k).cpu().numpy(), err_msg=k)
    counts = ("samples", "density_lookups", "color_lookups", "emission_lookups", "tracking_steps", "skip_steps", "queued_rays", "density_fetches")
    for c in counts:
        assert getattr(sa, c) == getattr(sb, c), c
    b.ctx.close()
    del b
    ob = oracle_binding.OracleBinding(sd)
    ob.render(1)
    assert np.isfinite(bufs["accum"]).all() and ob.accum.max() > 0
    np.testing.assert_array_equal(bufs["depth"], ob.depth)
    np.testing.assert_array_equal(bufs["raw"][:, 3], ob.raw[:, 3])
    assert sa.samples == ob.stats.samples == 1920 * 1080
    for c in ("density_lookups", "tracking_steps", "skip_steps"):
        assert getattr(sa, c) == getattr(ob.stats, c), (c, getattr(sa, c), getattr(ob.stats, c))
    assert sa.tracking_steps > sa.density_lookups > 0
    e = rel_l2(bufs["accum"], ob.accum)
    assert e <= 1e-3, e


def test_config5_100_instances_4k_dof_sun_and_sky(pkg, monkeypatch):
    sd = pkg.scene.instanced_scene(3840, 2160, n=128, grid=10, aperture=2.0, sky=True)
    pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    # The open lens: one ground table and one dome per binary32 step of r across the lens disc, each variant gated at build time on real rays
    # through both paths (vpt_tail.hip sky_dir_table_verdict_kernel).  Round 5 (profiles/r05_c5_p99.txt): with the ground-hit GEOMETRY formed as the
    # strict side forms it (vpt_sky.h) the per-sample evaluation lands on the oracle's own binary32 ground points -- WITHOUT the tables the per-pixel
    # 99th percentile after two iterations fell from 1.79e-3 to 1.84e-4 (worst pixel 4.8e-3 -> 1.05e-3, image 1.9e-4 -> 4.8e-5): asserted first, with
    # 2.7x of margin.
    monkeypatch.setenv("VPT_NO_DIR_TABLE", "1")
    _compare(pkg, sd, 2, caches=("cam_table",), p99=5e-4, worst=3e-3, tol=2e-4)
    monkeypatch.delenv("VPT_NO_DIR_TABLE")
    # ... and WITH tables and domes -- smooth caches of a function that is noisy from ray to ray: ~2 % of the ground hits find their binary32 ground point one
    # step (0.5 m) above the ground and come out ~5e-3 different (the reference's noise; a cache returns its mean, 1e-4 of the radiance).  After TWO
    # iterations 4 % of the ground pixels hold one such sample at half weight: the 99th percentile sat INSIDE that population (1.98e-3 / 1.99e-3 in rounds
    # 4 / 5 against the 2e-3 bound -- it measured the population's share, not an error).  Four iterations dilute a flipped sample to ~1.2e-3: measured
    # median 8.2e-5, 95th percentile 8.2e-4, 99th 1.12e-3 -- the same 2e-3 bound holds with 44 % of margin, and the median is held to 2e-4.
    # (resolved_samples: since round 5 the tracer and raygen resolve the samples from the dome of their origin's variant behind the open lens too)
    e, st = _compare(pkg, sd, 4, caches=("sky_dome", "cam_table", "dir_table", "resolved_samples"), median=2e-4)
    assert st.color_lookups > 0 and st.density_lookups > 2 * st.tracking_steps       # several instances per leaf and step
