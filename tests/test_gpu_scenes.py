"""Parity of the HIP path with the oracle on the scene features of BASELINE configs 3 and 5 at
sizes the oracle finishes in seconds: emission grids + blackbody LUT (estimate_emission,
render_kernel.cu:1275), colour grids (sum_color :931), many overlapping instances over the octree
(get_quadrant :1102, sum_density :1003), thin-lens DOF (camera.h:131), point lights over instances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_L2_TOL = 1e-3
REL_L2_TIGHT = 5e-6


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(1e-30, np.sqrt((b ** 2).sum())))


def _check(pkg, sd, spp, counts=("density_lookups", "color_lookups", "emission_lookups", "tracking_steps", "skip_steps")):
    import oracle_binding
    hb = pkg.scene.HipBinding(sd, device=0)
    ob = oracle_binding.OracleBinding(sd)
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
    np.testing.assert_allclose(hb.depth.cpu().numpy(), ob.depth, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(hb.raw.cpu().numpy()[:, 3], ob.raw[:, 3], rtol=1e-6, atol=1e-6)
    st = hb.ctx.stats()
    assert st.samples == ob.stats.samples == sd.width * sd.height * spp
    for c in counts:
        assert getattr(st, c) == getattr(ob.stats, c), c
    return st


def test_fireball_emission_parity(pkg):
    """config 3 shape: heat grid -> blackbody LUT, emission march after every interaction."""
    sd = pkg.scene.fireball_scene(128, 96, n=40)
    st = _check(pkg, sd, 3)
    assert st.emission_lookups > 0


def test_emission_scale_without_emission_grid_still_walks(pkg):
    """emission_scale > 0 on a scene with no heat grid: estimate_emission still marches (and
    consumes random numbers), adding nothing (render_kernel.cu:1285, :1802)."""
    sd = pkg.scene.dragon_scene(96, 64, "sun")
    sd.kp.emission_scale = 1.0
    _check(pkg, sd, 2)


def test_instanced_colour_dof_parity(pkg):
    """config 5 shape: 16 rotated instances of one coloured-smoke grid, octree with empty nodes,
    overlapping instance lists, aperture > 0."""
    sd = pkg.scene.instanced_scene(128, 96, n=20, grid=4, aperture=0.5)
    st = _check(pkg, sd, 3)
    assert st.color_lookups > 0 and st.skip_steps > 0
    assert st.density_lookups > st.tracking_steps      # several instances per look-up position


def test_instanced_with_point_lights_and_anisotropy(pkg):
    sd = pkg.scene.instanced_scene(96, 64, n=16, grid=3, aperture=0.0)
    from ctypes import c_float
    c = pkg.scene
    for k in range(3):
        pl = c.PointLight()
        pl.pos = c.f3(np.array([10.0 * (k - 1), 12.0, 4.0 * k], np.float32))
        pl.color = c.Float3(1.0, 0.8 - 0.2 * k, 0.5 + 0.2 * k)
        pl.power = 60.0
        sd.lights.append(pl)
    sd.kp.phase_g1 = 0.6
    sd.kp.volume_depth = 3
    sd.kp.ray_depth = 4
    sd.kp.density_mult = 2.0
    _check(pkg, sd, 2)


def test_sphere_in_view_and_viz_dof(pkg):
    """the reference sphere inside the frame: sphere bounce (:1807-1834), BLACK shadow rays
    (:1160), depth from the sphere hit (:1885)."""
    sd = pkg.scene.dragon_scene(128, 96, "sun")
    g = pkg.scene.load_golden("dragon_dense.npz")
    sd.sphere.center = pkg.scene.Float3(6.0, 1.5, 4.0)
    sd.sphere.radius = 1.2
    sd.sphere.color = pkg.scene.Float3(0.7, 0.6, 0.5)
    sd.sphere.roughness = 0.4
    _check(pkg, sd, 3)


def test_two_different_files_take_the_general_instance_path(pkg):
    """instances of DIFFERENT grids (per-instance descriptors differ in more than the transform):
    the compact single-file layout must not be used; density-only dragon + coloured smoke + a heat grid"""
    S = pkg.scene
    sd = S.dragon_scene(128, 96, "sun")
    dens, cd = S.smoke_grids(20)
    heat = (dens * dens).astype(np.float32)
    vdb = S.make_gpu_vdb(dens, (0, 0, 0), (19, 19, 19), S._grid_matrix(dens.shape, 0.25, centre=(9.0, 3.0, 5.0)), 0.25, emission=heat, color=cd)
    sd.volumes.append((vdb, dens, heat, cd))
    sd.kp.emission_scale = 0.5
    cam, _, _ = S.frame_camera(pkg.load_library(), [v[0] for v in sd.volumes], 128, 96)
    sd.camera = cam
    st = _check(pkg, sd, 3)
    assert st.color_lookups > 0 and st.emission_lookups > 0


def test_axis_aligned_instances_and_their_edge_shell(pkg):
    """Instances that are NOT rotated (round-1 advisor finding): Bounds() covers index space [bmin, bmax], the look-up accepts
    [bmin, bmin + dim] (render_kernel.cu:997 with dim = bmax - bmin + 1, gpu_vdb.cpp:453-455), so a point in the one-voxel shell
    past bmax is outside an instance's AABB yet still summed by the reference whenever the instance is in the leaf's list.  The
    refined per-sub-cell candidate lists must keep such an instance.  Coarse 8^3 grids (the shell is an eighth of the instance),
    every voxel non-zero, boxes abutting and overlapping: sums, look-up counts, depth and alpha equal the oracle's, which walks
    the leaf lists as the reference does."""
    n = 8
    rng = np.random.default_rng(3)
    dens = (0.2 + 0.8 * rng.random((n, n, n), dtype=np.float32)).astype(np.float32)
    cd = np.ones((n, n, n, 4), np.float32)
    cd[..., :3] = rng.random((n, n, n, 3), dtype=np.float32)
    for spacing in (8.0, 9.5):
        sd = pkg.scene.instanced_scene(112, 80, n=n, grid=4, aperture=0.0, rotate=False, spacing=spacing, grids=(dens, cd))
        sd.kp.density_mult = 0.5
        st = _check(pkg, sd, 3)
        assert st.color_lookups > 0
