"""Scenes on which the oracle is pinned against the reference's own kernel (oracle/_ref) -- TEST INFRASTRUCTURE.

Shared by tests/test_oracle_vs_ref.py (live comparison where oracle/_ref/libvptref.so exists) and
tests/golden/make_ref_golden.py (writes tests/golden/ref_golden.npz from the compiled reference, so that the pin also
holds where /root/reference does not exist).  Every case covers a different part of volume_rt_kernel:
"""
import ctypes as C

import numpy as np

from oracle_binding import pkg
from ref_binding import attach_synthetic_atmosphere

S = pkg.scene


def _dragon(cfg, w=96, h=54, **kw):
    def make():
        sd = S.dragon_scene(w, h, cfg)
        for k, v in kw.items():
            setattr(sd.kp, k, v)
        return sd
    return make


def _dragon_dof():
    sd = S.dragon_scene(64, 36, "sun")
    cam, _, _ = S.frame_camera(pkg.host.load_library(), [sd.volumes[0][0]], 64, 36, aperture=3.0)
    sd.camera = cam
    sd.camera.viz_dof = 1
    return sd


def _cloud(integrator, sky):
    def make():
        sd = S.cloud_scene(48, 32, shape=(38, 22, 32), env=(64, 32), integrator=integrator)
        if sky:
            sd.kp.environment_type = 0
            sd.env_map = None
            sd.env_cdf = pkg.host.env_cdf_build(sd.kp)
        return sd
    return make


def _two_files():
    """two different grids (one with emission, one with colour) + a point light: the multi-file path"""
    sd = S.fireball_scene(56, 40, n=32)
    other = S.instanced_scene(56, 40, n=20, grid=2)
    sd.volumes += other.volumes
    vols = [v for v, _, _, _ in sd.volumes]
    sd.camera, center, dist = S.frame_camera(pkg.host.load_library(), vols, 56, 40)
    pl = pkg.abi.PointLight()
    pl.pos = S.f3(center + np.array([0, dist, 0], np.float32))
    pl.color = pkg.abi.Float3(1.0, 0.8, 0.6)
    pl.power = float(dist * dist)
    sd.lights.append(pl)
    return sd


def _sphere_in_view():
    """the reference sphere inside the frame: sphere bounce, BLACK shadow rays, depth from the sphere hit"""
    sd = S.dragon_scene(96, 72, "sun")
    sd.sphere.center = S.Float3(6.0, 1.5, 4.0)
    sd.sphere.radius = 1.2
    sd.sphere.color = S.Float3(0.7, 0.6, 0.5)
    sd.sphere.roughness = 0.4
    return sd


def _vol_three_lights():
    """vol_integrator: 2 point lights + sun + HDRI, anisotropic phase: every branch of uniform_sample_one_light"""
    sd = S.dragon_scene(96, 72, "c2")
    sd.kp.integrator = 1
    sd.kp.environment_type = 1
    sd.env_map = S.hdri_map(64, 32)
    sd.kp.phase_g1 = 0.4
    sd.kp.ray_depth = 8
    sd.kp.density_mult = 3.0
    for k in range(2):
        pl = S.PointLight()
        pl.pos = S.f3(np.array([2.0 + 5.0 * k, 8.0, 5.0], np.float32))
        pl.color = S.Float3(1.0, 0.7, 0.4 + 0.5 * k)
        pl.power = 40.0
        sd.lights.append(pl)
    return sd


def _vol_emission_sphere():
    """vol_integrator over the fireball (emission at every interaction) with the sphere in the way"""
    sd = S.fireball_scene(64, 48, n=32)
    sd.kp.integrator = 1
    sd.kp.ray_depth = 5
    sd.kp.emission_scale = 0.7
    sd.sphere.center = S.Float3(4.0, 2.0, 3.0)
    sd.sphere.radius = 1.5
    return sd


def _multi_bounce():
    """volume_depth 3: several delta-tracking walks per outer bounce with HG scattering in between"""
    sd = S.dragon_scene(96, 54, "sun")
    sd.kp.volume_depth = 3
    sd.kp.phase_g1 = -0.3
    sd.kp.density_mult = 4.0
    sd.kp.ray_depth = 4
    return sd


def _xform_soup():
    """the reference's second asset (assets/dragon_with_xform.vdb: rotated + sheared AffineMap, 141x99x63) under
    non-default everything: coloured albedo / extinction, tr_depth, density_mult, energy_inject, sun position,
    exposure, a wide lens"""
    lib = pkg.host.load_library()
    g = S.load_golden("dragon_xform_dense.npz")
    sd = S.dragon_scene(80, 60, "c2")
    vdb = S.make_gpu_vdb(g["density"], g["bbox_min"], g["bbox_max"], g["matrix"], g["voxel_size"])
    sd.volumes = [(vdb, np.ascontiguousarray(g["density"], np.float32), None, None)]
    sd.camera, _, _ = S.frame_camera(lib, [vdb], 80, 60, fov=40.0, aperture=1.0)
    kp = sd.kp
    kp.albedo = S.Float3(0.9, 0.7, 0.5)
    kp.extinction = S.Float3(1.0, 1.2, 1.5)
    kp.tr_depth = 0.6
    kp.density_mult = 2.5
    kp.energy_inject = 1.3
    kp.azimuth = 250.0
    kp.elevation = 12.0
    kp.exposure_scale = 1.7
    kp.sun_color = S.Float3(1.0, 0.85, 0.7)
    kp.sun_mult = 2.0
    kp.sky_mult = 0.6
    kp.ray_depth = 7
    return sd


def _cloud_vol_dof():
    """vol_integrator behind an open lens (untraced samples carry their own origin)"""
    sd = S.cloud_scene(48, 32, shape=(38, 22, 32), env=(64, 32), integrator=1)
    vols = [v for v, _, _, _ in sd.volumes]
    sd.camera, _, _ = S.frame_camera(pkg.host.load_library(), vols, 48, 32, aperture=4.0)
    return sd


def _camera(sd, lookfrom, lookat, fov, w, h):
    cam = pkg.abi.Camera()
    lib = pkg.host.load_library()
    lib.vpt_camera_default(C.byref(cam))
    lib.vpt_camera_update(C.byref(cam), S.Float3(*lookfrom), S.Float3(*lookat), S.Float3(0, 1, 0), fov, float(w) / float(h), 0.0)
    sd.camera = cam


def _camera_inside():
    """camera INSIDE the root box: AABB::Intersect's "origin inside => tmin := tmax" rule moves every primary ray to the
    box's far side before the first walk, so the reference renders the background only (a quirk worth pinning)"""
    sd = S.dragon_scene(48, 32, "c2")
    _camera(sd, (5.0, 2.5, 5.0), (4.0, 2.0, 4.6), 70.0, 48, 32)
    return sd


def _three_point_lights():
    """close camera, strong forward scattering, three point lights: the light_budget loop runs its 11 Tr walks and adds
    Le for the last three (estimate_point_light :1445-1475)"""
    sd = S.dragon_scene(72, 48, "sun")
    _camera(sd, (10.5, 6.5, 9.0), (5.0, 2.5, 5.0), 45.0, 72, 48)
    sd.kp.phase_g1 = 0.85
    sd.kp.density_mult = 6.0
    sd.kp.ray_depth = 5
    for k in range(3):
        pl = S.PointLight()
        pl.pos = S.f3(np.array([1.0 + 3.0 * k, 6.0 - k, 2.0 + 2.0 * k], np.float32))
        pl.color = S.Float3(0.3 + 0.3 * k, 1.0, 1.0 - 0.3 * k)
        pl.power = 25.0
        sd.lights.append(pl)
    return sd


def _sphere_through_volume():
    """the reference sphere INTERSECTING the volume: walks that end on the sphere (obj = 2 inside sample()), shadow rays
    blocked by it, rough bounce"""
    sd = S.dragon_scene(80, 56, "c2")
    sd.sphere.center = S.Float3(4.6, 2.2, 4.8)
    sd.sphere.radius = 0.9
    sd.sphere.color = S.Float3(0.9, 0.4, 0.3)
    sd.sphere.roughness = 0.85
    sd.kp.density_mult = 3.0
    return sd


def _thin_and_extreme():
    """energy_inject 0 (beta turns black: isBlack break), tr_depth 3, exposure 0.2, ray_depth 1, nearly empty medium"""
    sd = S.dragon_scene(64, 40, "c2")
    sd.kp.energy_inject = 0.0
    sd.kp.tr_depth = 3.0
    sd.kp.exposure_scale = 0.2
    sd.kp.ray_depth = 1
    sd.kp.density_mult = 0.25
    sd.kp.phase_g1 = -0.95
    return sd


# name -> (scene factory, iterations)
CASES = {
    "dragon_point_light": (_dragon("c1"), 3),                       # point-light NEE, direct_integrator
    "dragon_sun": (_dragon("sun"), 3),                              # sun NEE through the atmosphere tables
    "dragon_sun_sky": (_dragon("c2"), 3),                           # + sky tail (GetSkyRadiance) on every path
    "dragon_hg_forward": (_dragon("sun", phase_g1=0.7, ray_depth=6, tr_depth=0.5), 2),
    "dragon_max_interactions": (_dragon("sun", max_interactions=2), 4),   # iterations >= max_interactions stop accumulating
    "dragon_no_render": (_dragon("sun", render=0), 2),
    "dragon_dof_viz": (_dragon_dof, 2),                             # thin lens (van der Corput rejection) + viz_dof
    "fireball_emission": (lambda: S.fireball_scene(64, 40, n=48), 2),    # estimate_emission + blackbody table
    "instanced_colour": (lambda: S.instanced_scene(64, 40, n=24, grid=4), 2),  # 16 instances, Cd grids, octree with empty nodes
    "two_files_point_light": (_two_files, 2),
    "cloud_vol_hdri": (_cloud(1, False), 2),                        # vol_integrator, HDRI background
    "cloud_direct_hdri": (_cloud(0, False), 2),                     # direct_integrator on the HDRI
    "cloud_vol_sky_cdf": (_cloud(1, True), 2),                      # vol_integrator, estimate_sky over the CDF tables
    "dragon_sphere_in_view": (_sphere_in_view, 3),
    "dragon_vol_three_lights": (_vol_three_lights, 2),
    "fireball_vol_emission_sphere": (_vol_emission_sphere, 2),
    "dragon_multi_bounce_back_scatter": (_multi_bounce, 2),
    "dragon_xform_parameter_soup": (_xform_soup, 2),
    "cloud_vol_hdri_open_lens": (_cloud_vol_dof, 2),
    "dragon_camera_inside_root_box": (_camera_inside, 2),
    "dragon_three_point_lights_forward": (_three_point_lights, 2),
    "dragon_sphere_through_volume": (_sphere_through_volume, 3),
    "dragon_black_beta_extremes": (_thin_and_extreme, 2),
}

BUFFERS = ("accum", "depth", "raw", "display", "blue_noise")


def build(name):
    make, iters = CASES[name]
    sd = make()
    attach_synthetic_atmosphere(sd)
    return sd, iters
