"""The screen-space bounds behind the renderer's never-traced pixel mask (csrc/vpt_host.hip: project_box) against brute force:
for random closed-lens cameras and boxes, every jittered primary ray that hits the box with the reference's own slab test
(AABB::Intersect, bvh/AABB.h:182-205, restated in numpy with binary32 operands) must come from a pixel inside the projected rectangle
grown by the renderer's three-pixel margin -- and the rectangle must not be much larger than the hits it bounds.  Host only."""
import ctypes as C

import numpy as np
import pytest


def _camera(pkg, lookfrom, lookat, fov, aspect):
    from vpt_amd.abi import Camera, Float3
    cam = Camera()
    lib = pkg.load_library()
    lib.vpt_camera_default(C.byref(cam))
    lib.vpt_camera_update(C.byref(cam), Float3(*lookfrom), Float3(*lookat), Float3(0, 1, 0), float(fov), float(aspect), 0.0)
    return cam


def _v(a):
    return np.array([a.x, a.y, a.z], np.float32)


def _hits(cam, lo, hi, W, H, rng, n=60000):
    """pixels (x, y) of random jittered rays that pass AABB::Intersect, in binary32 like the kernel"""
    x = rng.integers(0, W, n); y = rng.integers(0, H, n)
    jx = rng.random(n, dtype=np.float32); jy = rng.random(n, dtype=np.float32)
    # pixel corners too: the extremes of a footprint
    jx[: n // 4] = np.float32(rng.integers(0, 2, n // 4)); jy[: n // 4] = np.float32(rng.integers(0, 2, n // 4)) * np.float32(0.99999994)
    jx[: n // 4] *= np.float32(0.99999994)
    u = ((x.astype(np.float32) + jx) / np.float32(W))[:, None]; v = ((y.astype(np.float32) + jy) / np.float32(H))[:, None]
    o = _v(cam.origin)
    d = _v(cam.lower_left_corner) + u * _v(cam.horizontal) + v * _v(cam.vertical) - o
    d = (d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]).astype(np.float32)
    with np.errstate(all="ignore"):
        inv = (np.float32(1.0) / d).astype(np.float32)
        t1 = (lo - o) * inv; t2 = (hi - o) * inv
    tmin = np.minimum(t1, t2).max(1); tmax = np.maximum(t1, t2).min(1)
    hit = (tmax > 0) & ~(tmin > tmax)
    return x[hit], y[hit]


def test_projected_box_bounds_every_hit(pkg):
    lib = pkg.load_library()
    lib.vpt_test_project_box.argtypes = [C.c_void_p, C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), C.c_int, C.c_int, C.POINTER(C.c_float * 4)]
    rng = np.random.default_rng(3)
    W, H = 320, 180
    bounded = tight = 0
    for trial in range(60):
        lo = rng.uniform(-8, 4, 3).astype(np.float32)
        hi = (lo + rng.uniform(0.5, 9, 3)).astype(np.float32)
        centre = (lo + hi) / 2
        dist = float(rng.choice([3.0, 12.0, 40.0, 400.0, 20000.0]))
        dirn = rng.standard_normal(3); dirn /= np.linalg.norm(dirn)
        lookfrom = centre + dirn * (np.linalg.norm(hi - lo) / 2 + dist)
        lookat = centre + rng.uniform(-1, 1, 3) * (hi - lo) * float(rng.choice([0.2, 1.5]))       # sometimes the box is half off the frame
        cam = _camera(pkg, lookfrom, lookat, float(rng.choice([1.0, 20.0, 55.0, 100.0])), W / H)
        rect = (C.c_float * 4)()
        rc = lib.vpt_test_project_box(C.byref(cam), (C.c_float * 3)(*lo), (C.c_float * 3)(*hi), W, H, C.byref(rect))
        if rc != 0:
            continue                                            # a corner behind the camera plane: the renderer skips nothing
        bounded += 1
        hx, hy = _hits(cam, lo, hi, W, H, rng)
        if hx.size == 0:
            continue
        m = 3.0
        assert hx.min() + 1 >= rect[0] - m and hx.max() <= rect[2] + m, (trial, hx.min(), hx.max(), list(rect))
        assert hy.min() + 1 >= rect[1] - m and hy.max() <= rect[3] + m, (trial, hy.min(), hy.max(), list(rect))
        # ... and not a loose bound: the hits reach close to each side of the rectangle that lies inside the frame (a side is touched by one
        # corner of the box: the random rays find its neighbourhood, not the corner itself)
        for lo_side, hit_edge in ((rect[0], hx.min()), (rect[1], hy.min())):
            if lo_side > 2:
                tight += 1
                assert hit_edge - lo_side <= 0.1 * max(W, H), (trial, lo_side, hit_edge)
    assert bounded >= 30 and tight >= 10


def test_projection_refuses_boxes_that_reach_behind_the_camera(pkg):
    lib = pkg.load_library()
    lib.vpt_test_project_box.argtypes = [C.c_void_p, C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), C.c_int, C.c_int, C.POINTER(C.c_float * 4)]
    cam = _camera(pkg, (0.0, 0.0, 0.0), (0.0, 0.0, -5.0), 45.0, 16 / 9)          # inside the box
    rect = (C.c_float * 4)()
    assert lib.vpt_test_project_box(C.byref(cam), (C.c_float * 3)(-1, -1, -1), (C.c_float * 3)(1, 1, 1), 320, 180, C.byref(rect)) != 0
    assert lib.vpt_test_project_box(C.byref(cam), (C.c_float * 3)(-1, -1, 3), (C.c_float * 3)(1, 1, 5), 320, 180, C.byref(rect)) != 0    # all behind
    assert lib.vpt_test_project_box(C.byref(cam), (C.c_float * 3)(-1, -1, -9), (C.c_float * 3)(1, 1, -7), 320, 180, C.byref(rect)) == 0
    assert rect[0] < 160 < rect[2] and rect[1] < 90 < rect[3]


def _sphere_intersect_f32(o, d, c, r):
    """sphere::intersect + find_discr (geometry/geometry.h:46-70, 114-137) on binary32 arrays, operation by operation -> hit mask"""
    f = np.float32
    ox, oy, oz = (o[:, 0] - c[0]).astype(f), (o[:, 1] - c[1]).astype(f), (o[:, 2] - c[2]).astype(f)
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    A = ((dx * dx).astype(f) + (dy * dy).astype(f)).astype(f) + (dz * dz).astype(f)
    B = f(2) * (((dx * ox).astype(f) + (dy * oy).astype(f)).astype(f) + (dz * oz).astype(f))
    C = (((ox * ox).astype(f) + (oy * oy).astype(f)).astype(f) + (oz * oz).astype(f)) - f(r) * f(r)
    with np.errstate(all="ignore"):
        discr = (B * B).astype(f) - ((f(4) * A).astype(f) * C).astype(f)
        sq = np.sqrt(discr).astype(f)
        q = np.where(B < 0, f(-0.5) * (B - sq), f(-0.5) * (B + sq)).astype(f)
        x1 = (q / A).astype(f); x2 = (C / q).astype(f)
        tmin = np.minimum(x1, x2); tmax = np.maximum(x1, x2)
    hit = (discr >= 0) & ((tmin >= 0) | (tmax >= 0)) & (B != 0)
    return hit


def test_inflated_sphere_bounds_every_reported_hit(pkg):
    """sphere::intersect decides with a binary32 discriminant that cancels catastrophically when the sphere is far away: it reports hits for
    rays that pass the centre at many radii (a unit sphere 19 km away: out to ~10 units).  The never-traced mask's sphere test
    (csrc/vpt_cull.h: sphere_may_hit) must say "may hit" for every ray the reference's arithmetic reports as a hit."""
    lib = pkg.load_library()
    lib.vpt_test_sphere_may_hit.argtypes = [C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), C.c_float, C.POINTER(C.c_float * 4)]
    rng = np.random.default_rng(17)
    reported = beyond_radius = 0
    for D, r in ((19000.0, 1.0), (1000.0, 1.0), (300.0, 0.5), (50.0, 2.0), (6.0, 1.5), (2.0e5, 3.0)):
        c = np.array([0.0, 1000.0, 0.0], np.float32)
        dirn = rng.standard_normal(3); dirn /= np.linalg.norm(dirn)
        o = (c + dirn * D).astype(np.float32)
        n = 200000
        # rays aimed at points within a few inflated radii of the centre
        spread = 6.0 * np.sqrt(r * r + 64 * 1.19e-7 * D * D)
        target = c + rng.uniform(-spread, spread, (n, 3))
        d = (target - o); d = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32)
        hit = _sphere_intersect_f32(np.tile(o, (n, 1)), d, c, r)
        p = np.linalg.norm(np.cross((c - o).astype(np.float64), d.astype(np.float64)), axis=1)      # true distance of the centre from each ray
        reported += int(hit.sum())
        beyond_radius += int((hit & (p > 1.5 * r)).sum())
        sph = (C.c_float * 4)(c[0], c[1], c[2], r)
        for i in np.flatnonzero(hit)[:4000]:
            assert lib.vpt_test_sphere_may_hit((C.c_float * 3)(*o), (C.c_float * 3)(*d[i]), 0.0, sph) == 1, (D, r, p[i])
        # ... and the test does cull: rays passing at 20 inflated radii, or pointing away, cannot hit
        far = np.flatnonzero(p > 3.0 * spread)
        for i in far[:200]:
            assert not hit[i]
        away = (-d[0]).astype(np.float32)
        assert lib.vpt_test_sphere_may_hit((C.c_float * 3)(*o), (C.c_float * 3)(*away), 0.0, sph) == 0
        wide = (c + np.array([40.0 * spread, 0, 0]) - o); wide = (wide / np.linalg.norm(wide)).astype(np.float32)
        if 40.0 * spread < 0.5 * D:
            assert lib.vpt_test_sphere_may_hit((C.c_float * 3)(*o), (C.c_float * 3)(*wide), 0.0, sph) == 0
    assert reported > 1000 and beyond_radius > 20        # the cancellation is real: hits well outside the geometric sphere were seen
