"""Minimal OpenVDB (format 224) WRITER -- TEST INFRASTRUCTURE for the product's C++ reader
(csrc/vpt_io.hip): builds small files with known content so the reader is exercised without the
reference's assets (which do not travel to the GPU box).  Layout: SURVEY.md appendix A.
One root child per 4096^3 block, Internal5 -> Internal4 -> Leaf, optional ZIP compression and
active-mask value compression, float and vec3s grids, UniformScale / Affine maps, active tiles."""
import struct
import zlib

import numpy as np

ZIP, ACTIVE_MASK = 1, 2


def _s(x):
    b = x.encode("latin-1")
    return struct.pack("<I", len(b)) + b


def _bits(mask):
    return np.packbits(np.asarray(mask, bool), bitorder="little").tobytes()


def _values(vals, vmask, flags, background, half=False):
    """compressed_values(N): vals (N, C) float32, vmask (N,) bool.  half: the value block is binary16 (io::HalfWriter);
    the inactive value stays a 4-byte float, truncated to half precision, as OpenVDB writes it"""
    vals = np.asarray(vals, np.float32)
    n = vals.shape[0]
    out = b""
    if flags & ACTIVE_MASK:
        inactive = vals[~vmask]
        if inactive.size and not (inactive == np.asarray(background, np.float32)).all():
            # one non-background inactive value: metadata 2 (NO_MASK_AND_ONE_INACTIVE_VAL)
            iv = inactive[0]
            assert (inactive == iv).all(), "writer supports at most one inactive value"
            if half:
                iv = np.asarray(iv, np.float16).astype(np.float32)
            out += struct.pack("<b", 2) + np.asarray(iv, "<f4").tobytes()
        else:
            out += struct.pack("<b", 0)
        data = vals[vmask]
    else:
        out += struct.pack("<b", 6)
        data = vals
    raw = np.ascontiguousarray(data, "<f2" if half else "<f4").tobytes()
    if flags & ZIP:
        if len(raw) <= 64:                       # small buffers are stored raw with a negative size
            out += struct.pack("<q", -len(raw)) + raw
        else:
            z = zlib.compress(raw)
            out += struct.pack("<q", len(z)) + z
    else:
        out += raw
    return out


def _grid_bytes(leaves, tiles3, ncomp, background, flags, map_type, map_values, meta, half=False):
    """leaves: {(ox,oy,oz): (vals[512,C], mask[512])}, origins multiples of 8 inside ONE 4096^3 root
    child at origin (0,0,0) ... ; tiles3: {(ox,oy,oz): val} active 8^3 tiles."""
    bg = np.zeros(ncomp, np.float32) + np.asarray(background, np.float32)
    head = struct.pack("<I", flags)
    head += struct.pack("<I", len(meta))
    for k, (t, v) in meta.items():
        head += _s(k) + _s(t) + struct.pack("<I", len(v)) + v
    head += _s(map_type) + np.asarray(map_values, "<f8").tobytes()
    topo = struct.pack("<i", 1) + bg.astype("<f4").tobytes()
    roots = sorted({tuple((np.array(o) >> 12 << 12).tolist()) for o in list(leaves) + list(tiles3)})
    topo += struct.pack("<II", 0, len(roots))
    bufs = b""
    for ro in roots:
        topo += struct.pack("<3i", *ro)
        cm5 = np.zeros(32768, bool); vm5 = np.zeros(32768, bool)
        kids5 = {}
        for o in list(leaves) + list(tiles3):
            if tuple((np.array(o) >> 12 << 12).tolist()) != ro:
                continue
            rel = np.array(o) - np.array(ro)
            i = ((rel[0] >> 7) << 10) | ((rel[1] >> 7) << 5) | (rel[2] >> 7)
            cm5[i] = True
            kids5.setdefault(i, []).append(o)
        topo += _bits(cm5) + _bits(vm5) + _values(np.tile(bg, (32768, 1)), vm5, flags, bg, half)
        for i in sorted(kids5):
            o5 = np.array(ro) + np.array([(i >> 10) << 7, ((i >> 5) & 31) << 7, (i & 31) << 7])
            cm4 = np.zeros(4096, bool); vm4 = np.zeros(4096, bool)
            vals4 = np.tile(bg, (4096, 1))
            order = {}
            for o in kids5[i]:
                rel = np.array(o) - o5
                j = ((rel[0] >> 3) << 8) | ((rel[1] >> 3) << 4) | (rel[2] >> 3)
                if o in leaves:
                    cm4[j] = True
                    order[j] = o
                else:
                    vm4[j] = True
                    vals4[j] = tiles3[o]
            topo += _bits(cm4) + _bits(vm4) + _values(vals4, vm4, flags, bg, half)
            for j in sorted(order):
                vals, mask = leaves[order[j]]
                topo += _bits(mask)
                bufs += _bits(mask) + _values(np.asarray(vals, np.float32).reshape(512, ncomp), np.asarray(mask, bool), flags, bg, half)
    return head + topo, bufs


def write_vdb(path, grids):
    """grids: list of dict(name, type ('float'|'vec3s'), leaves, tiles3, background, flags, map_type, map_values, half)"""
    out = struct.pack("<qIII", 0x56444220, 224, 5, 2) + b"\x01" + b"0" * 36
    out += struct.pack("<I", 0)                                 # file metadata
    out += struct.pack("<I", len(grids))
    blobs = []
    for g in grids:
        ncomp = 3 if g["type"] == "vec3s" else 1
        half = bool(g.get("half", False))
        meta_half = bool(g.get("meta_half", half))               # (tests can make the metadatum contradict the descriptor suffix)
        meta = {"class": ("string", b"fog volume"), "is_saved_as_half_float": ("bool", b"\x01" if meta_half else b"\x00")}
        a, b = _grid_bytes(g["leaves"], g.get("tiles3", {}), ncomp, g.get("background", 0.0), g.get("flags", 0),
                           g.get("map_type", "UniformScaleMap"), g["map_values"], meta, half)
        blobs.append((g, a, b))
    pos = len(out)
    def tname(g):                                               # GridDescriptor: the type name carries the half-float suffix
        return "Tree_%s_5_4_3" % g["type"] + ("_HalfFloat" if g.get("half") else "")
    for g, a, b in blobs:                                       # descriptor sizes first
        pos += len(_s(g["name"])) + len(_s(tname(g))) + len(_s("")) + 24
    descs = b""
    body = b""
    cur = len(out)
    # descriptors are interleaved with the grids in real files: name/type/parent/offsets then the grid
    res = out
    for g, a, b in blobs:
        d = _s(g["name"]) + _s(tname(g)) + _s("")
        grid_pos = len(res) + len(d) + 24
        block_pos = grid_pos + len(a)
        end_pos = block_pos + len(b)
        res += d + struct.pack("<qqq", grid_pos, block_pos, end_pos) + a + b
    with open(path, "wb") as f:
        f.write(res)


def uniform_scale(s):
    return [s, s, s, s, s, s, 1 / s, 1 / s, 1 / s, 1 / s ** 2, 1 / s ** 2, 1 / s ** 2, 0.5 / s, 0.5 / s, 0.5 / s]


def dense_to_leaves(dense, bbox_min):
    """[z, y, x] dense array -> {leaf origin: (vals[512,1], mask[512])} with active = non-zero"""
    nz, ny, nx = dense.shape
    lo = np.asarray(bbox_min, np.int64)
    leaves = {}
    o0 = (lo >> 3) << 3
    hi = lo + np.array([nx, ny, nz]) - 1
    for ox in range(o0[0], hi[0] + 1, 8):
        for oy in range(o0[1], hi[1] + 1, 8):
            for oz in range(o0[2], hi[2] + 1, 8):
                blk = np.zeros((8, 8, 8), np.float32)              # [x][y][z]
                xs = slice(max(ox, lo[0]), min(ox + 8, hi[0] + 1))
                ys = slice(max(oy, lo[1]), min(oy + 8, hi[1] + 1))
                zs = slice(max(oz, lo[2]), min(oz + 8, hi[2] + 1))
                sub = dense[zs.start - lo[2]:zs.stop - lo[2], ys.start - lo[1]:ys.stop - lo[1], xs.start - lo[0]:xs.stop - lo[0]]
                blk[xs.start - ox:xs.stop - ox, ys.start - oy:ys.stop - oy, zs.start - oz:zs.stop - oz] = sub.transpose(2, 1, 0)
                if (blk != 0).any():
                    v = blk.reshape(512, 1)                        # n = (x << 6) | (y << 3) | z
                    leaves[(int(ox), int(oy), int(oz))] = (v, (v[:, 0] != 0))
    return leaves
