"""Minimal pure-Python OpenVDB (file format 222-224) reader -- TEST INFRASTRUCTURE.

Used (a) by make_fixtures.py to turn the reference's assets into the committed golden
fixtures and (b) as the independent checker of the product's C++ reader.  It follows the
on-disk layout documented in SURVEY.md appendix A (OpenVDB is a third-party dependency
of the reference, `vcpkg.json:5`; the reference's only use of it is
source/gpu_vdb/gpu_vdb.cpp:133-250,414-459: read grid -> evalActiveVoxelBoundingBox ->
copyToDense(LayoutXYZ) -> max/min density -> index-to-world matrix).

Handles: Tree_float_5_4_3 and Tree_vec3s_5_4_3, compression flags ZIP(1) |
ACTIVE_MASK(2) | BLOSC(4), UniformScaleMap / UniformScaleTranslateMap / ScaleMap /
ScaleTranslateMap / AffineMap, half-float storage is rejected.
"""
import ctypes
import struct
import zlib

import numpy as np

_blosc = None


def _blosc_lib():
    global _blosc
    if _blosc is None:
        for name in ("/opt/conda/lib/libblosc.so", "libblosc.so.1", "libblosc.so"):
            try:
                _blosc = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _blosc is None:
            raise RuntimeError("libblosc not found")
    return _blosc


class _R:
    def __init__(self, buf):
        self.b = buf
        self.p = 0

    def take(self, n):
        v = self.b[self.p:self.p + n]
        if len(v) != n:
            raise EOFError("truncated vdb")
        self.p += n
        return v

    def u8(self): return self.take(1)[0]
    def i8(self): return struct.unpack("<b", self.take(1))[0]
    def u32(self): return struct.unpack("<I", self.take(4))[0]
    def i32(self): return struct.unpack("<i", self.take(4))[0]
    def i64(self): return struct.unpack("<q", self.take(8))[0]
    def f32(self): return struct.unpack("<f", self.take(4))[0]
    def f64s(self, n): return struct.unpack("<%dd" % n, self.take(8 * n))
    def string(self): return bytes(self.take(self.u32())).decode("latin-1")


COMPRESS_ZIP, COMPRESS_ACTIVE_MASK, COMPRESS_BLOSC = 1, 2, 4


def _read_raw(r, nbytes_out, flags):
    """the framed (possibly compressed) byte block of compressed_values()."""
    if flags & COMPRESS_BLOSC:
        n = r.i64()
        if n <= 0:
            return bytes(r.take(-n))
        src = bytes(r.take(n))
        dst = ctypes.create_string_buffer(nbytes_out)
        got = _blosc_lib().blosc_decompress_ctx(src, dst, ctypes.c_size_t(nbytes_out), 1)
        if got != nbytes_out:
            raise ValueError("blosc: expected %d bytes, got %d" % (nbytes_out, got))
        return dst.raw
    if flags & COMPRESS_ZIP:
        n = r.i64()
        if n <= 0:
            return bytes(r.take(-n))
        out = zlib.decompress(bytes(r.take(n)))
        if len(out) != nbytes_out:
            raise ValueError("zip size mismatch")
        return out
    return bytes(r.take(nbytes_out))


def _read_values(r, n, value_mask_bits, flags, dtype, ncomp, background):
    """compressed_values(N) of appendix A -> float32 array (n, ncomp)."""
    metadata = r.i8()
    inactive0 = np.array(background, dtype=np.float32)
    inactive1 = np.array(background, dtype=np.float32)
    if metadata == 1:
        inactive0 = -inactive0
    if metadata in (2, 4, 5):
        inactive0 = np.frombuffer(r.take(4 * ncomp), dtype="<f4").copy()
    if metadata == 5:
        inactive1 = np.frombuffer(r.take(4 * ncomp), dtype="<f4").copy()
    sel = None
    if metadata in (3, 4, 5):
        sel = np.unpackbits(np.frombuffer(r.take(n // 8), dtype=np.uint8), bitorder="little").astype(bool)
    if (flags & COMPRESS_ACTIVE_MASK) and metadata != 6:
        count = int(value_mask_bits.sum())
    else:
        count = n
    raw = _read_raw(r, count * 4 * ncomp, flags)
    vals = np.frombuffer(raw, dtype="<f4").reshape(count, ncomp)
    if count == n:
        return vals.astype(np.float32, copy=True)
    out = np.empty((n, ncomp), dtype=np.float32)
    out[value_mask_bits] = vals
    inactive = ~value_mask_bits
    if sel is None:
        out[inactive] = inactive0
    else:
        out[inactive & ~sel] = inactive0
        out[inactive & sel] = inactive1
    return out


def _bits(r, nbits):
    return np.unpackbits(np.frombuffer(r.take(nbits // 8), dtype=np.uint8), bitorder="little").astype(bool)


class Grid:
    """Sparse grid as a list of (origin, 8x8x8xC values, 8^3 active mask) leaves + active tiles."""

    def __init__(self):
        self.name = ""
        self.grid_type = ""
        self.meta = {}
        self.map_type = ""
        self.matrix = np.eye(4)          # OpenVDB Mat4d, row-vector convention
        self.voxel_size = 1.0
        self.background = None
        self.ncomp = 1
        self.leaves = []                 # (origin(3,), values(512,C), mask(512,))
        self.tiles = []                  # (origin(3,), log2dim, value(C,))

    def active_bbox(self):
        lo = np.array([2**31 - 1] * 3, dtype=np.int64)
        hi = -lo
        for org, vals, mask in self.leaves:
            if not mask.any():
                continue
            m = mask.reshape(8, 8, 8)
            idx = np.argwhere(m)
            lo = np.minimum(lo, org + idx.min(0))
            hi = np.maximum(hi, org + idx.max(0))
        for org, lg, val in self.tiles:
            lo = np.minimum(lo, org)
            hi = np.maximum(hi, org + (1 << lg) - 1)
        return lo, hi

    def active_voxel_count(self):
        return int(sum(int(m.sum()) for _, _, m in self.leaves) + sum((1 << lg) ** 3 for _, lg, _ in self.tiles))

    def to_dense(self):
        """copyToDense over the active-voxel bbox, LayoutXYZ (x fastest): returns
        (array[z, y, x, C], bbox_min, bbox_max)."""
        lo, hi = self.active_bbox()
        dim = (hi - lo + 1).astype(int)
        out = np.empty((dim[0], dim[1], dim[2], self.ncomp), dtype=np.float32)
        out[...] = np.asarray(self.background, dtype=np.float32)
        for org, lg, val in self.tiles:
            s = np.maximum(org - lo, 0)
            e = np.minimum(org + (1 << lg) - lo, dim)
            out[s[0]:e[0], s[1]:e[1], s[2]:e[2]] = val
        for org, vals, mask in self.leaves:
            o = org - lo
            blk = vals.reshape(8, 8, 8, self.ncomp)
            s = np.maximum(o, 0)
            e = np.minimum(o + 8, dim)
            if (e <= s).any():
                continue
            out[s[0]:e[0], s[1]:e[1], s[2]:e[2]] = blk[s[0] - o[0]:e[0] - o[0], s[1] - o[1]:e[1] - o[1], s[2] - o[2]:e[2] - o[2]]
        # (x, y, z, C) -> memory order z, y, x (x fastest)
        return np.ascontiguousarray(out.transpose(2, 1, 0, 3)), lo, hi


def _read_meta_value(r, typ):
    size = r.u32()
    raw = r.take(size)
    if typ == "string":
        return bytes(raw).decode("latin-1")
    if typ == "bool":
        return bool(raw[0])
    if typ == "int32":
        return struct.unpack("<i", raw)[0]
    if typ == "int64":
        return struct.unpack("<q", raw)[0]
    if typ == "float":
        return struct.unpack("<f", raw)[0]
    if typ == "double":
        return struct.unpack("<d", raw)[0]
    if typ == "vec3i":
        return struct.unpack("<3i", raw)
    if typ == "vec3s":
        return struct.unpack("<3f", raw)
    if typ == "vec3d":
        return struct.unpack("<3d", raw)
    return bytes(raw)


def _read_transform(r, g):
    g.map_type = r.string()
    m = np.eye(4)
    if g.map_type in ("UniformScaleMap", "ScaleMap"):
        v = r.f64s(15)
        m[0, 0], m[1, 1], m[2, 2] = v[0], v[1], v[2]
        g.voxel_size = v[3]
    elif g.map_type in ("UniformScaleTranslateMap", "ScaleTranslateMap"):
        v = r.f64s(18)
        tr, sc = v[0:3], v[3:6]
        m[0, 0], m[1, 1], m[2, 2] = sc
        m[3, 0:3] = tr
        g.voxel_size = v[6]
    elif g.map_type == "TranslationMap":
        v = r.f64s(3)
        m[3, 0:3] = v
    elif g.map_type == "AffineMap":
        m = np.array(r.f64s(16)).reshape(4, 4)
        g.voxel_size = float(np.linalg.norm(m[0, 0:3]))
    else:
        raise ValueError("unsupported map " + g.map_type)
    g.matrix = m


def _read_grid(r, buf, name, gtype, grid_pos, block_pos, file_version):
    g = Grid()
    g.name, g.grid_type = name, gtype
    if gtype.startswith("Tree_float_5_4_3"):
        g.ncomp = 1
    elif gtype.startswith("Tree_vec3s_5_4_3"):
        g.ncomp = 3
    else:
        raise ValueError("unsupported grid type " + gtype)
    if gtype.endswith("_HalfFloat"):
        raise ValueError("half-float grids unsupported")
    r.p = grid_pos
    flags = r.u32()
    for _ in range(r.u32()):
        mname = r.string()
        mtype = r.string()
        g.meta[mname] = _read_meta_value(r, mtype)
    if g.meta.get("is_saved_as_half_float"):
        raise ValueError("half-float grids unsupported")
    _read_transform(r, g)
    # ---- topology
    buffer_count = r.i32()
    assert buffer_count == 1
    g.background = np.frombuffer(r.take(4 * g.ncomp), dtype="<f4").copy()
    num_tiles = r.u32()
    num_children = r.u32()
    for _ in range(num_tiles):
        xyz = np.array(struct.unpack("<3i", r.take(12)), dtype=np.int64)
        val = np.frombuffer(r.take(4 * g.ncomp), dtype="<f4").copy()
        active = r.u8()
        if active:
            g.tiles.append((xyz, 12, val))
    leaf_origins = []
    for _ in range(num_children):
        org5 = np.array(struct.unpack("<3i", r.take(12)), dtype=np.int64)
        cm5 = _bits(r, 32768)
        vm5 = _bits(r, 32768)
        vals5 = _read_values(r, 32768, vm5, flags, "f4", g.ncomp, g.background)
        for i in np.flatnonzero(vm5 & ~cm5):
            o = org5 + np.array([(i >> 10) << 7, ((i >> 5) & 31) << 7, (i & 31) << 7])
            g.tiles.append((o, 7, vals5[i]))
        for i in np.flatnonzero(cm5):
            org4 = org5 + np.array([(i >> 10) << 7, ((i >> 5) & 31) << 7, (i & 31) << 7])
            cm4 = _bits(r, 4096)
            vm4 = _bits(r, 4096)
            vals4 = _read_values(r, 4096, vm4, flags, "f4", g.ncomp, g.background)
            for j in np.flatnonzero(vm4 & ~cm4):
                o = org4 + np.array([(j >> 8) << 3, ((j >> 4) & 15) << 3, (j & 15) << 3])
                g.tiles.append((o, 3, vals4[j]))
            for j in np.flatnonzero(cm4):
                o = org4 + np.array([(j >> 8) << 3, ((j >> 4) & 15) << 3, (j & 15) << 3])
                _bits(r, 512)             # leaf value mask (topology pass)
                leaf_origins.append(o)
    # ---- buffers
    r.p = block_pos
    for o in leaf_origins:
        vm = _bits(r, 512)
        vals = _read_values(r, 512, vm, flags, "f4", g.ncomp, g.background)
        g.leaves.append((o, vals, vm))
    g.end_pos = r.p
    return g


def read_vdb(path):
    """returns {grid_name: Grid}"""
    with open(path, "rb") as f:
        buf = f.read()
    r = _R(memoryview(buf))
    magic = r.i64()
    if magic != 0x56444220:
        raise ValueError("not a VDB file")
    version = r.u32()
    if version < 222:
        raise ValueError("file version %d < 222 unsupported" % version)
    r.u32(); r.u32()                  # library major / minor
    has_offsets = r.u8()
    r.take(36)                        # uuid
    for _ in range(r.u32()):
        r.string(); t = r.string(); _read_meta_value(r, t)
    grids = {}
    ngrids = r.u32()
    descs = []
    for _ in range(ngrids):
        name = r.string()
        gtype = r.string()
        parent = r.string()
        grid_pos, block_pos, end_pos = r.i64(), r.i64(), r.i64()
        descs.append((name, gtype, grid_pos, block_pos, end_pos))
        if not has_offsets:
            raise ValueError("files without grid offsets unsupported")
        r.p = end_pos
    for name, gtype, gp, bp, ep in descs:
        g = _read_grid(r, buf, name, gtype, gp, bp, version)
        if g.end_pos != ep:
            raise ValueError("grid %s: parsed to %d, descriptor says %d" % (name, g.end_pos, ep))
        grids[name] = g
    return grids
