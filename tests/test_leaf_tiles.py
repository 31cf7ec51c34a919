"""The tile map the renderer refines its never-traced pixel mask with (csrc/vpt_caches.hip: build_leaf_tiles) against brute force.
Claim: a pixel whose 8x8 tile is NOT marked sends no primary ray through any NON-EMPTY octree leaf -- so every ray of it crosses empty nodes only and
ends as a ray that misses the box.  (a) random closed-lens cameras, root boxes and occupancy sets: every jittered ray that passes the reference's
slab test (AABB::Intersect, bvh/AABB.h:182-205, binary32 operands) against some non-empty leaf comes from a marked tile, and the map is not much
larger than those hits; (b) the occupancy-bit convention (path = 64 c1 + 8 c2 + c3, c = x high | y LOW << 1 | z high << 2) against the ORACLE's octree
on the dragon: points sampled along rays of unmarked tiles never lie in a leaf that lists a volume.  Host only."""
import ctypes as C

import numpy as np
import pytest

from test_cull_bounds import _camera, _v


def _leaf_boxes(lo, hi):
    """[512, 2, 3] leaf boxes of the root [lo, hi] in the product's path order (binary64 halving, as the host does it)"""
    out = np.zeros((512, 2, 3))
    for path in range(512):
        l = np.array(lo, np.float64); h = np.array(hi, np.float64)
        for level in range(3):
            c = (path >> (6 - 3 * level)) & 7
            high = ((c & 1) != 0, (c & 2) == 0, (c & 4) != 0)
            for a in range(3):
                mid = (l[a] + h[a]) * 0.5
                if high[a]:
                    l[a] = mid
                else:
                    h[a] = mid
        out[path, 0] = l; out[path, 1] = h
    return out


def _occ_words(nonempty_paths):
    occ = (C.c_uint * 19)()
    for p in nonempty_paths:
        occ[(96 + p) >> 5] |= 1 << (p & 31)
    return occ


def _rays(cam, W, H, rng, n):
    x = rng.integers(0, W, n); y = rng.integers(0, H, n)
    jx = rng.random(n, dtype=np.float32); jy = rng.random(n, dtype=np.float32)
    jx[: n // 4] = np.float32(rng.integers(0, 2, n // 4)) * np.float32(0.99999994)       # pixel corners too: the extremes of a footprint
    jy[: n // 4] = np.float32(rng.integers(0, 2, n // 4)) * np.float32(0.99999994)
    u = ((x.astype(np.float32) + jx) / np.float32(W))[:, None]; v = ((y.astype(np.float32) + jy) / np.float32(H))[:, None]
    o = _v(cam.origin)
    d = _v(cam.lower_left_corner) + u * _v(cam.horizontal) + v * _v(cam.vertical) - o
    d = (d / np.sqrt((d * d).sum(1, dtype=np.float32))[:, None]).astype(np.float32)
    return x, y, o, d


def _tiles(pkg, cam, lo, hi, occ, W, H, margin=3.0):
    lib = pkg.load_library()
    tw, th = (W + 7) // 8, (H + 7) // 8
    buf = (C.c_ubyte * (tw * th))()
    f3 = C.c_float * 3
    lib.vpt_test_leaf_tiles.argtypes = [C.c_void_p, f3, f3, C.c_uint * 19, C.c_int, C.c_int, C.c_float, C.c_ubyte * (tw * th)]
    rc = lib.vpt_test_leaf_tiles(C.byref(cam), f3(*lo), f3(*hi), occ, W, H, margin, buf)
    return rc, np.frombuffer(buf, np.uint8).reshape(th, tw).copy()


def test_unmarked_tiles_see_no_nonempty_leaf(pkg):
    rng = np.random.default_rng(20260927)
    W, H = 320, 180
    cases = 0
    for trial in range(40):
        c = rng.uniform(-20, 20, 3)
        half = rng.uniform(2, 25, 3)
        lo = (c - half).astype(np.float32); hi = (c + half).astype(np.float32)
        dist = rng.uniform(2.5, 8.0) * float(half.max())
        dirn = rng.normal(size=3); dirn /= np.linalg.norm(dirn)
        look_at = c + rng.uniform(-0.5, 0.5, 3) * half
        cam = _camera(pkg, tuple(look_at + dirn * dist), tuple(look_at), rng.uniform(25, 70), W / H)
        k = int(rng.integers(1, 120))
        nonempty = sorted(set(int(p) for p in rng.integers(0, 512, k)))
        if trial % 5 == 0:                                    # a compact blob, like a real asset in its padded box
            nonempty = [p for p in range(512) if ((p >> 6) & 7) in (0, 3) and (p & 7) in (1, 5, 6)]
        rc, tiles = _tiles(pkg, cam, lo, hi, _occ_words(nonempty), W, H)
        if rc != 0:
            continue                                          # a leaf corner behind the camera plane: the renderer refines nothing
        cases += 1
        boxes = _leaf_boxes(lo, hi)[nonempty].astype(np.float32)       # binary32 operands, like the kernel's slab test
        x, y, o, d = _rays(cam, W, H, rng, 30000)
        with np.errstate(all="ignore"):
            inv = (np.float32(1.0) / d).astype(np.float32)
            t1 = (boxes[None, :, 0, :] - o) * inv[:, None, :]; t2 = (boxes[None, :, 1, :] - o) * inv[:, None, :]
        tmin = np.minimum(t1, t2).max(2); tmax = np.maximum(t1, t2).min(2)
        hit = ((tmax > 0) & ~(tmin > tmax)).any(1)
        assert hit.any()
        assert tiles[y[hit] >> 3, x[hit] >> 3].all(), "a ray through a non-empty leaf comes from an unmarked tile (trial %d)" % trial
        # ... and the map is tight: the tiles the rays hit from, grown by one tile (margin 3 px + tile granularity), cover what is marked
        seen = np.zeros_like(tiles, bool)
        seen[y[hit] >> 3, x[hit] >> 3] = True
        grown = seen.copy()
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                grown |= np.roll(np.roll(seen, dy, 0), dx, 1)
        marked = tiles.astype(bool)
        assert (marked & ~grown).sum() <= 0.25 * marked.sum() + 8, (int((marked & ~grown).sum()), int(marked.sum()))
    assert cases >= 25


def test_occupancy_convention_against_the_oracles_octree(pkg, orc):
    import oracle_binding
    W, H = 960, 540
    sd = pkg.scene.dragon_scene(W, H, "c2")
    ob = oracle_binding.OracleBinding(sd)
    info = oracle_binding.OctreeInfo()
    assert orc.orc_octree_info_get(ob.volumes, 1, C.byref(info)) == 0
    lo = np.array([info.root_pmin.x, info.root_pmin.y, info.root_pmin.z], np.float32)
    hi = np.array([info.root_pmax.x, info.root_pmax.y, info.root_pmax.z], np.float32)
    boxes = _leaf_boxes(lo, hi)
    nv = C.c_int(0)

    def volumes_at(p):
        orc.orc_octree_locate(ob.volumes, 1, pkg.abi.Float3(float(p[0]), float(p[1]), float(p[2])), C.byref(nv))
        return nv.value
    nonempty = [p for p in range(512) if volumes_at(boxes[p].mean(0)) > 0]
    assert len(nonempty) == info.nonempty[2]                 # the oracle's own count of non-empty level-3 nodes
    assert 0 < len(nonempty) < 512                           # the dragon fills a fraction of its padded box
    rc, tiles = _tiles(pkg, sd.camera, lo, hi, _occ_words(nonempty), W, H)
    assert rc == 0 and 0 < tiles.sum() < tiles.size
    rng = np.random.default_rng(7)
    x, y, o, d = _rays(sd.camera, W, H, rng, 60000)
    unmarked = tiles[y >> 3, x >> 3] == 0
    with np.errstate(all="ignore"):
        inv = 1.0 / d.astype(np.float64)
        t1 = (lo - o) * inv; t2 = (hi - o) * inv
    tmin = np.minimum(t1, t2).max(1); tmax = np.maximum(t1, t2).min(1)
    through_root = unmarked & (tmax > np.maximum(tmin, 0))
    assert through_root.sum() > 200                          # rays that hit the box but, by the map, only empty nodes
    for i in np.nonzero(through_root)[0][:600]:
        for t in np.linspace(max(tmin[i], 0.0), tmax[i], 48)[1:-1]:
            assert volumes_at(o + d[i].astype(np.float64) * t) == 0, (int(x[i]), int(y[i]))
