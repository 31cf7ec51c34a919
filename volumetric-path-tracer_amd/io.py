"""ctypes binding of include/vpt_io.h: the native (C++) readers/writers for the formats either side
of the hot path -- OpenVDB files, `.ins` instance files, BN0.bmp, the 256x1 EXR look-up tables,
Radiance HDR maps, PFM / PPM output (reference: gpu_vdb.cpp:105-472, main.cpp:980-1102,
fileIO.cpp, hdr_loader.h)."""
import ctypes as C

import numpy as np

from . import abi
from .abi import GpuVdb, Int3, PointLight
from .host import VptError, load_library


class Instance(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("rotation", C.c_double * 4), ("scale", C.c_double)]


IO_SYMBOLS = [
    "vpt_io_last_error", "vpt_io_free", "vpt_io_vdb_load", "vpt_io_vdb_free", "vpt_io_vdb_info", "vpt_io_vdb_grid",
    "vpt_io_vdb_stats", "vpt_io_vdb_upload", "vpt_io_ins_read", "vpt_io_ins_free", "vpt_io_ins_is_light_file",
    "vpt_io_ins_num_files", "vpt_io_ins_file_name", "vpt_io_ins_num_instances", "vpt_io_ins_instances",
    "vpt_io_ins_num_lights", "vpt_io_ins_lights", "vpt_io_load_bmp", "vpt_io_load_exr_rgb", "vpt_io_load_hdr",
    "vpt_io_write_pfm", "vpt_io_write_ppm", "vpt_io_write_png", "vpt_io_write_png_float",
]

_ready = False


def _lib():
    global _ready
    lib = load_library()
    if not _ready:
        vp, cp = C.c_void_p, C.c_char_p
        lib.vpt_io_last_error.restype = cp
        lib.vpt_io_free.argtypes = [vp]
        lib.vpt_io_free.restype = None
        lib.vpt_io_vdb_load.argtypes = [cp, cp, cp, cp, C.POINTER(vp)]
        lib.vpt_io_vdb_free.argtypes = [vp]
        lib.vpt_io_vdb_free.restype = None
        lib.vpt_io_vdb_info.argtypes = [vp, C.POINTER(GpuVdb)]
        lib.vpt_io_vdb_grid.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(Int3)]
        lib.vpt_io_vdb_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_longlong * 3)]
        lib.vpt_io_vdb_upload.argtypes = [vp, vp, C.POINTER(GpuVdb)]
        lib.vpt_io_ins_read.argtypes = [cp, C.POINTER(vp)]
        lib.vpt_io_ins_free.argtypes = [vp]
        lib.vpt_io_ins_free.restype = None
        for f in ("vpt_io_ins_is_light_file", "vpt_io_ins_num_files", "vpt_io_ins_num_lights"):
            getattr(lib, f).argtypes = [vp]
        lib.vpt_io_ins_file_name.argtypes = [vp, C.c_int]
        lib.vpt_io_ins_file_name.restype = cp
        lib.vpt_io_ins_num_instances.argtypes = [vp, C.c_int]
        lib.vpt_io_ins_instances.argtypes = [vp, C.c_int]
        lib.vpt_io_ins_instances.restype = C.POINTER(Instance)
        lib.vpt_io_ins_lights.argtypes = [vp]
        lib.vpt_io_ins_lights.restype = C.POINTER(PointLight)
        for f in ("vpt_io_load_bmp", "vpt_io_load_exr_rgb", "vpt_io_load_hdr"):
            getattr(lib, f).argtypes = [cp, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.vpt_io_write_pfm.argtypes = [cp, vp, C.c_int, C.c_int, C.c_int]
        lib.vpt_io_write_ppm.argtypes = [cp, vp, C.c_int, C.c_int]
        lib.vpt_io_write_png.argtypes = [cp, vp, C.c_int, C.c_int, C.c_int]
        lib.vpt_io_write_png_float.argtypes = [cp, vp, C.c_int, C.c_int, C.c_int]
        _ready = True
    return lib


def _chk(rc, what):
    if rc != 0:
        raise VptError("%s -> %s: %s" % (what, abi.E_NAMES.get(rc, rc), _lib().vpt_io_last_error().decode()))


class VdbFile:
    """GPU_VDB::loadVDB(filename, "density", "heat", "Cd") without OpenVDB."""

    def __init__(self, path, density="density", emission="heat", color="Cd"):
        lib = _lib()
        h = C.c_void_p()
        _chk(lib.vpt_io_vdb_load(path.encode(), density.encode(), (emission or "").encode(), (color or "").encode(), C.byref(h)), "vpt_io_vdb_load")
        self.h = h
        self.info = GpuVdb()
        _chk(lib.vpt_io_vdb_info(h, C.byref(self.info)), "vpt_io_vdb_info")

    def grid(self, which):
        """0 density [z,y,x], 1 emission [z,y,x], 2 colour [z,y,x,4]; None when absent (copy)."""
        lib = _lib()
        p = C.POINTER(C.c_float)()
        dim = Int3()
        rc = lib.vpt_io_vdb_grid(self.h, which, C.byref(p), C.byref(dim))
        if rc == -5:
            return None
        _chk(rc, "vpt_io_vdb_grid")
        shape = (dim.z, dim.y, dim.x, 4) if which == 2 else (dim.z, dim.y, dim.x)
        return np.ctypeslib.as_array(p, shape=shape).copy()

    def stats(self, which=0):
        out = (C.c_longlong * 3)()
        _chk(_lib().vpt_io_vdb_stats(self.h, which, C.byref(out)), "vpt_io_vdb_stats")
        return {"leaves": out[0], "active_voxels": out[1], "active_tiles": out[2]}

    def upload(self, ctx):
        out = GpuVdb()
        _chk(_lib().vpt_io_vdb_upload(ctx.h, self.h, C.byref(out)), "vpt_io_vdb_upload")
        return out

    def close(self):
        if getattr(self, "h", None):
            _lib().vpt_io_vdb_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_ins(path):
    """-> {"lights": [PointLight...]} or {"files": [(name, [Instance...])...]}"""
    lib = _lib()
    h = C.c_void_p()
    _chk(lib.vpt_io_ins_read(path.encode(), C.byref(h)), "vpt_io_ins_read")
    try:
        if lib.vpt_io_ins_is_light_file(h):
            n = lib.vpt_io_ins_num_lights(h)
            p = lib.vpt_io_ins_lights(h)
            return {"lights": [PointLight.from_buffer_copy(p[i]) for i in range(n)]}
        files = []
        for f in range(lib.vpt_io_ins_num_files(h)):
            n = lib.vpt_io_ins_num_instances(h, f)
            p = lib.vpt_io_ins_instances(h, f)
            files.append((lib.vpt_io_ins_file_name(h, f).decode(), [Instance.from_buffer_copy(p[i]) for i in range(n)]))
        return {"files": files}
    finally:
        lib.vpt_io_ins_free(h)


def _load_image(fn, path, channels):
    lib = _lib()
    p = C.POINTER(C.c_float)()
    w, h = C.c_int(), C.c_int()
    _chk(getattr(lib, fn)(path.encode(), C.byref(p), C.byref(w), C.byref(h)), fn)
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, channels)).copy()
    finally:
        lib.vpt_io_free(p)


def load_bmp(path):
    return _load_image("vpt_io_load_bmp", path, 3)


def load_exr_rgb(path):
    return _load_image("vpt_io_load_exr_rgb", path, 3)


def load_hdr(path):
    return _load_image("vpt_io_load_hdr", path, 4)


def write_pfm(path, pixels, width, height):
    a = np.ascontiguousarray(pixels, np.float32).reshape(height, width, -1)
    _chk(_lib().vpt_io_write_pfm(path.encode(), a.ctypes.data_as(C.c_void_p), a.shape[2], width, height), "vpt_io_write_pfm")


def write_ppm(path, display, width, height):
    a = np.ascontiguousarray(display).view(np.uint32).reshape(height, width)
    _chk(_lib().vpt_io_write_ppm(path.encode(), a.ctypes.data_as(C.c_void_p), width, height), "vpt_io_write_ppm")


def write_png(path, display, width, height, with_alpha=False):
    """8-bit PNG of the 0xffRRGGBB display buffer (the reference's save_texture_png(uint32_t*), fileIO.cpp:140-154)"""
    a = np.ascontiguousarray(display).view(np.uint32).reshape(height, width)
    _chk(_lib().vpt_io_write_png(path.encode(), a.ctypes.data_as(C.c_void_p), width, height, 1 if with_alpha else 0), "vpt_io_write_png")


def write_png_float(path, pixels, width, height):
    """8-bit PNG of a float3 / float4 image, clamped to [0, 1] (OpenImageIO's FLOAT -> UINT8 conversion, fileIO.cpp:110-138)"""
    a = np.ascontiguousarray(pixels, np.float32).reshape(height, width, -1)
    _chk(_lib().vpt_io_write_png_float(path.encode(), a.ctypes.data_as(C.c_void_p), a.shape[2], width, height), "vpt_io_write_png_float")
