"""ctypes binding of the CPU oracle (oracle/_build/liborc.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
abi = pkg.abi

ORC_PATH = os.path.join(ROOT, "oracle", "_build", "liborc.so")


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in ("samples", "density_lookups", "color_lookups", "emission_lookups",
                                             "tracking_steps", "skip_steps", "rng_draws", "max_draws_per_sample")]


class OctreeInfo(C.Structure):
    _fields_ = [("root_pmin", abi.Float3), ("root_pmax", abi.Float3), ("max_extinction", C.c_float),
                ("min_extinction", C.c_float), ("nonempty", C.c_int * 3), ("total_nodes", C.c_int)]


_orc = None


def load_oracle():
    global _orc
    if _orc is not None:
        return _orc
    if not os.path.exists(ORC_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    o = C.CDLL(ORC_PATH)
    vp = C.c_void_p
    o.orc_texture_create.argtypes = [C.POINTER(abi.TextureDesc), vp]
    o.orc_texture_create.restype = abi.vpt_texture_t
    o.orc_texture_destroy.argtypes = [abi.vpt_texture_t]
    o.orc_texture_destroy.restype = None
    o.orc_texture_sample.argtypes = [abi.vpt_texture_t, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float * 4)]
    o.orc_texture_sample.restype = None
    o.orc_set_volume_tex_weights.argtypes = [C.c_int]
    o.orc_set_volume_tex_weights.restype = C.c_int
    o.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint * 4), C.POINTER(C.c_uint * 2), C.POINTER(C.c_uint * 4)]
    o.orc_philox4x32_10.restype = None
    o.orc_curand_uniform_stream.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_int, vp]
    o.orc_curand_uniform_stream.restype = None
    for f in ("orc_det_logf", "orc_det_sinf", "orc_det_cosf"):
        getattr(o, f).argtypes = [C.c_float]
        getattr(o, f).restype = C.c_float
    o.orc_octree_info_get.argtypes = [C.POINTER(abi.GpuVdb), C.c_int, C.POINTER(OctreeInfo)]
    o.orc_octree_locate.argtypes = [C.POINTER(abi.GpuVdb), C.c_int, abi.Float3, C.POINTER(C.c_int)]
    o.orc_density_at.argtypes = [C.POINTER(abi.GpuVdb), C.c_int, abi.Float3]
    o.orc_density_at.restype = C.c_float
    render_args = [C.POINTER(abi.Camera), C.POINTER(abi.LightList), C.POINTER(abi.GpuVdb), C.c_int, C.POINTER(abi.Sphere),
                   C.POINTER(abi.AtmosphereParameters), C.POINTER(abi.KernelParams)]
    o.orc_render.argtypes = render_args + [C.c_uint, C.c_uint, C.c_int, C.POINTER(OrcStats)]
    o.orc_render_subset.argtypes = render_args + [C.c_uint, C.c_uint, C.c_int, C.c_uint, C.POINTER(OrcStats)]
    o.orc_sample_pixel.argtypes = render_args + [C.c_int, C.c_int, C.POINTER(C.c_float * 5)]
    _orc = o
    return o


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleBinding:
    """Host-memory twin of scene.HipBinding: same SceneDesc, numpy buffers, oracle textures."""

    def __init__(self, sd):
        self.o = load_oracle()
        self.sd = sd
        self._keep = []
        vols = []
        for vdb, dens, emis, col in sd.volumes:
            v = abi.GpuVdb.from_buffer_copy(vdb)
            v.vdb_info.density_texture = self.texture(dens, 1)
            if emis is not None:
                v.vdb_info.emission_texture = self.texture(emis, 1)
            if col is not None:
                v.vdb_info.color_texture = self.texture(col, 4)
            vols.append(v)
        self.volumes = (abi.GpuVdb * len(vols))(*vols)
        n = sd.width * sd.height
        self.accum = np.zeros((n, 3), np.float32)
        self.cost = np.zeros((n, 3), np.float32)
        self.depth = np.zeros(n, np.float32)
        self.raw = np.zeros((n, 4), np.float32)
        self.display = np.zeros(n, np.uint32)
        self.blue_noise = sd.blue_noise.copy()
        self.emission_lut = sd.emission_lut.copy()
        self.density_color_lut = sd.density_color_lut.copy()
        kp = abi.KernelParams.from_buffer_copy(sd.kp)
        kp.accum_buffer = self.accum.ctypes.data
        kp.cost_buffer = self.cost.ctypes.data
        kp.depth_buffer = self.depth.ctypes.data
        kp.raw_buffer = self.raw.ctypes.data
        kp.display_buffer = self.display.ctypes.data
        kp.blue_noise_buffer = self.blue_noise.ctypes.data
        kp.emission_texture = self.emission_lut.ctypes.data
        kp.density_color_texture = self.density_color_lut.ctypes.data
        self.kp = kp
        self.atmosphere = abi.AtmosphereParameters.from_buffer_copy(sd.atmosphere)
        if sd.atm_luts is not None:
            L = sd.atm_luts
            wc = (abi.ADDR_WRAP, abi.ADDR_CLAMP, abi.ADDR_CLAMP)
            self.atmosphere.transmittance_texture = self.texture(L["transmittance"], 4, address=wc)
            self.atmosphere.irradiance_texture = self.texture(L["irradiance"], 4, address=wc)
            self.atmosphere.scattering_texture = self.texture(L["scattering"], 4)
            self.atmosphere.single_mie_scattering_texture = self.texture(L["single_mie"], 4)
        if sd.env_map is not None:
            kp.env_tex = self.texture(sd.env_map, 4, address=(abi.ADDR_WRAP, abi.ADDR_CLAMP, abi.ADDR_CLAMP))
        if sd.env_cdf is not None:
            pkg.scene.bind_env_cdf(kp, sd.env_cdf, self.texture)
        self._lights_arr = (abi.PointLight * max(1, len(sd.lights)))(*sd.lights)
        self.lights = abi.LightList(len(sd.lights), C.cast(self._lights_arr, C.POINTER(abi.PointLight)))
        self.stats = OrcStats()

    def texture(self, data, channels, normalized=True, linear=True, address=(abi.ADDR_CLAMP,) * 3):
        a = np.ascontiguousarray(data, dtype=np.float32)
        self._keep.append(a)
        shape = a.shape[:-1] if channels == 4 else a.shape
        dims = list(shape)[::-1] + [1, 1]
        desc = abi.TextureDesc(dims[0], dims[1], dims[2], channels, int(normalized), int(linear), (C.c_int * 3)(*address))
        return self.o.orc_texture_create(C.byref(desc), _ptr(a))

    def render(self, iter_count, iter_stride=1, iteration=None, nthreads=0, pixel_step=1):
        """pixel_step > 1: only the pixels whose index is a multiple of it (orc_render_subset)"""
        if iteration is not None:
            self.kp.iteration = int(iteration)
        if nthreads <= 0:
            nthreads = os.cpu_count() or 1
        rc = self.o.orc_render_subset(C.byref(self.sd.camera), C.byref(self.lights), self.volumes, len(self.volumes), C.byref(self.sd.sphere),
                                      C.byref(self.atmosphere), C.byref(self.kp), int(iter_count), int(iter_stride), int(nthreads), int(pixel_step),
                                      C.byref(self.stats))
        if rc != 0:
            raise RuntimeError("orc_render -> %d" % rc)
        self.kp.iteration += int(iter_count) * int(iter_stride)

    def sample_pixel(self, x, y, iteration=0):
        kp = abi.KernelParams.from_buffer_copy(self.kp)
        kp.iteration = int(iteration)
        out = (C.c_float * 5)()
        rc = self.o.orc_sample_pixel(C.byref(self.sd.camera), C.byref(self.lights), self.volumes, len(self.volumes), C.byref(self.sd.sphere),
                                     C.byref(self.atmosphere), C.byref(kp), int(x), int(y), C.byref(out))
        if rc != 0:
            raise RuntimeError("orc_sample_pixel -> %d" % rc)
        return np.array(list(out), np.float32)
