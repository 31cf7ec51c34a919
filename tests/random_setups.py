"""Seeded random set-ups around the reference's dragon.vdb (tests/test_gpu_parity.py::test_random_setups_vs_oracle, tools/sky_random_probe.py)."""
import ctypes as C

import numpy as np


def dragon_setup(pkg, rs, case, sizes=((128, 113, 96), (72, 67, 54))):
    """-> (scene, w, h, sky, description).  Camera around (case 5: inside) the volume's box, the reference sphere from overlapping the box's edge to far away, sun direction,
    loop depths, density, phase g; the volume rotated about y in a third of the cases, point lights (+ sun) in a quarter, the procedural sky in a quarter."""
    from vpt_amd.abi import Float3
    lib = pkg.load_library()
    w, h = int(rs.choice(sizes[0])), int(rs.choice(sizes[1]))
    sky = case % 4 == 2
    sd = pkg.scene.dragon_scene(w, h, "c2" if sky else ("c1" if case % 4 == 3 else "sun"))
    if sky:
        pkg.atmosphere.attach_default_atmosphere(sd, device=0)
    if case % 4 == 3:
        sd.kp.sun_mult = 1.0
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
    eye = ctr + d * size * float(rs.uniform(0.2 if case == 5 else 1.3, 4.0))
    look = ctr + rs.uniform(-0.6, 0.6, 3) * half
    fov = float(rs.uniform(20.0, 70.0))
    lib.vpt_camera_update(C.byref(sd.camera), Float3(*[float(v) for v in eye]), Float3(*[float(v) for v in look]), Float3(0, 1, 0), fov, w / h, 0.0)
    sdir = rs.normal(size=3); sdir /= np.linalg.norm(sdir)
    sd.sphere.radius = float(rs.uniform(0.15, 0.6) if case % 2 else rs.uniform(0.8, 3.0))
    sc = ctr + sdir * (float(np.max(half)) + sd.sphere.radius * float(rs.uniform(0.8, 1.3) if case % 2 == 0 else rs.uniform(1.0, 4.0)))
    if sky:
        # the sky model's ground is y = 0 (render_kernel.cu:840: earth_center = (0, -bottom_radius, 0)); a bounce point below it is a view point INSIDE the planet, where
        # sample_atmosphere's binary32 geometry (radii of 6.36e6 with half-metre steps) is noise on either side -- keep the sphere above the ground
        sc[1] = max(sc[1], sd.sphere.radius + 0.5)
    sd.sphere.center = Float3(float(sc[0]), float(sc[1]), float(sc[2]))
    sd.kp.azimuth = float(rs.uniform(0.0, 360.0))
    sd.kp.elevation = float(rs.uniform(2.0, 88.0))
    sd.kp.ray_depth = int(rs.randint(1, 5))
    sd.kp.volume_depth = int(rs.randint(1, 4))
    sd.kp.density_mult = float(sd.kp.density_mult) * float(rs.choice([0.25, 1.0, 3.0]))
    sd.kp.phase_g1 = float(rs.uniform(-0.3, 0.85))
    desc = "%d x %d eye %s look %s fov %.1f sun az %.1f el %.1f sphere c %s r %.2f ray_depth %d volume_depth %d" % (
        w, h, np.round(eye, 2), np.round(look, 2), fov, sd.kp.azimuth, sd.kp.elevation, np.round(sc, 2), sd.sphere.radius, sd.kp.ray_depth, sd.kp.volume_depth)
    return sd, w, h, sky, desc


def random_view(pkg, rs, sd, w, h, aperture=0.0, inside=False, above_ground=False):
    """A seeded random camera around (inside=True: within) the union of the scene's volume bounds, a random sun, loop depths, density and phase g; the reference sphere near
    the volumes half of the time (above_ground: kept above y = 0, see dragon_setup).  -> description"""
    from vpt_amd.abi import Float3
    lib = pkg.load_library()
    lo_u, hi_u = np.full(3, np.inf), np.full(3, -np.inf)
    for vol in sd.volumes:
        lo, hi = Float3(), Float3()
        lib.vpt_gpu_vdb_bounds(C.byref(vol[0]), C.byref(lo), C.byref(hi))
        lo_u = np.minimum(lo_u, [lo.x, lo.y, lo.z]); hi_u = np.maximum(hi_u, [hi.x, hi.y, hi.z])
    ctr, half = (lo_u + hi_u) * 0.5, (hi_u - lo_u) * 0.5
    size = float(np.linalg.norm(half))
    d = rs.normal(size=3); d /= np.linalg.norm(d); d[1] = abs(d[1]) * 0.7
    eye = ctr + d * size * float(rs.uniform(0.1, 0.5) if inside else rs.uniform(1.2, 3.5))
    if above_ground:
        eye[1] = max(eye[1], 1.0)
    look = ctr + rs.uniform(-0.5, 0.5, 3) * half
    fov = float(rs.uniform(20.0, 65.0))
    lib.vpt_camera_update(C.byref(sd.camera), Float3(*[float(v) for v in eye]), Float3(*[float(v) for v in look]), Float3(0, 1, 0), fov, w / h, float(aperture))
    if rs.uniform() < 0.5:
        sdir = rs.normal(size=3); sdir /= np.linalg.norm(sdir)
        sd.sphere.radius = float(rs.uniform(0.05, 0.3) * size)
        sc = ctr + sdir * (float(np.max(half)) + sd.sphere.radius * float(rs.uniform(0.9, 2.5)))
        if above_ground:
            sc[1] = max(sc[1], sd.sphere.radius + 0.5)
        sd.sphere.center = Float3(float(sc[0]), float(sc[1]), float(sc[2]))
    sd.kp.azimuth = float(rs.uniform(0.0, 360.0))
    sd.kp.elevation = float(rs.uniform(2.0, 88.0))
    sd.kp.ray_depth = int(rs.randint(1, 4))
    sd.kp.volume_depth = int(rs.randint(1, 4))
    sd.kp.density_mult = float(sd.kp.density_mult) * float(rs.choice([0.5, 1.0, 2.0]))
    sd.kp.phase_g1 = float(rs.uniform(-0.3, 0.85))
    return "%d x %d eye %s look %s fov %.1f aperture %.2f sun az %.1f el %.1f ray_depth %d volume_depth %d" % (
        w, h, np.round(eye, 2), np.round(look, 2), fov, aperture, sd.kp.azimuth, sd.kp.elevation, sd.kp.ray_depth, sd.kp.volume_depth)
