"""Python host side above the C ABI (include/vpt_abi.h): ctypes bindings to
libvpt_hip.so plus a thin `Context` mirroring how the reference's main loop drives
`volume_rt_kernel` (reference source/main.cpp:1301-1303, 1313, 1350-1376, 1527-1546,
1822-1829).  PyTorch only supplies HBM buffers, the HIP stream and torch.distributed.

There is NO CPU fallback: if the HIP library is missing or no gfx950 device is visible
every render call raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .abi import (AtmosphereParameters, Camera, Float3, GpuVdb, KernelParams, LightList, PointLight,
                  RenderStats, Sphere, TextureDesc, vpt_texture_t)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvpt_hip.so")

# every symbol include/vpt_abi.h declares
ABI_SYMBOLS = [
    "vpt_create", "vpt_destroy", "vpt_last_error", "vpt_abi_version", "vpt_stream", "vpt_sync",
    "vpt_texture_create", "vpt_texture_create_device", "vpt_texture_destroy", "vpt_invalidate_sky_tables",
    "vpt_frame_ahead_invalidate", "vpt_set_frame_ahead",
    "vpt_scene_set_volumes", "vpt_scene_get_root", "vpt_scene_get_octree_stats",
    "vpt_render", "vpt_render_batch", "vpt_blue_noise_advance",
    "vpt_set_counting", "vpt_get_stats",
    "vpt_comm_unique_id", "vpt_comm_init_rank", "vpt_comm_destroy", "vpt_allreduce_accum", "vpt_resolve_display",
    "vpt_atmosphere_default_model", "vpt_atmosphere_model_options_default", "vpt_atmosphere_model", "vpt_atmosphere_precompute", "vpt_atmosphere_precompute_model", "vpt_atmosphere_read_lut",
    "vpt_env_cdf_build", "vpt_env_cdf_create",
    "vpt_camera_update", "vpt_camera_frame", "vpt_camera_default", "vpt_gpu_vdb_bounds", "vpt_instance_xform", "vpt_kernel_params_default",
]


class VptError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen libvpt_hip.so and declare the prototypes.  Raises if the library was not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("VPT_LIB_PATH") or LIB_PATH      # VPT_LIB_PATH: perf-experiment variants (build.py --variant)
    if not os.path.exists(p):
        raise VptError("%s not found: run `python volumetric-path-tracer_amd/build.py` (hipcc, gfx950)" % p)
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.vpt_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.vpt_create.restype = C.c_int
    lib.vpt_destroy.argtypes = [vp]
    lib.vpt_destroy.restype = None
    lib.vpt_last_error.argtypes = [vp]
    lib.vpt_last_error.restype = C.c_char_p
    lib.vpt_abi_version.restype = C.c_int
    lib.vpt_stream.argtypes = [vp]
    lib.vpt_stream.restype = vp
    lib.vpt_sync.argtypes = [vp]
    lib.vpt_texture_create.argtypes = [vp, C.POINTER(TextureDesc), vp, C.POINTER(vpt_texture_t)]
    lib.vpt_texture_create_device.argtypes = [vp, C.POINTER(TextureDesc), vp, C.POINTER(vpt_texture_t)]
    lib.vpt_texture_destroy.argtypes = [vp, vpt_texture_t]
    lib.vpt_invalidate_sky_tables.argtypes = [vp]
    lib.vpt_frame_ahead_invalidate.argtypes = [vp]
    lib.vpt_set_frame_ahead.argtypes = [vp, C.c_int]
    lib.vpt_scene_set_volumes.argtypes = [vp, C.POINTER(GpuVdb), C.c_int]
    lib.vpt_scene_get_root.argtypes = [vp, C.POINTER(Float3), C.POINTER(Float3), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.vpt_scene_get_octree_stats.argtypes = [vp, C.POINTER(C.c_int * 3)]
    render_args = [vp, C.POINTER(Camera), C.POINTER(LightList), C.POINTER(Sphere), C.POINTER(AtmosphereParameters), C.POINTER(KernelParams)]
    lib.vpt_render.argtypes = render_args + [vp]
    lib.vpt_render_batch.argtypes = render_args + [C.c_uint, C.c_uint, vp]
    lib.vpt_blue_noise_advance.argtypes = [vp, vp, C.c_uint, C.c_uint, vp]
    lib.vpt_set_counting.argtypes = [vp, C.c_int]
    lib.vpt_get_stats.argtypes = [vp, C.POINTER(RenderStats)]
    lib.vpt_comm_unique_id.argtypes = [vp]
    lib.vpt_comm_init_rank.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.vpt_comm_destroy.argtypes = [vp]
    lib.vpt_allreduce_accum.argtypes = [vp, vp, C.c_ulonglong, C.c_uint, vp]
    lib.vpt_resolve_display.argtypes = [vp, C.POINTER(KernelParams), vp]
    lib.vpt_camera_update.argtypes = [C.POINTER(Camera), Float3, Float3, Float3, C.c_float, C.c_float, C.c_float]
    lib.vpt_camera_update.restype = None
    lib.vpt_camera_frame.argtypes = [C.POINTER(Camera), C.POINTER(GpuVdb), C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(Float3), C.POINTER(C.c_float)]
    lib.vpt_camera_frame.restype = None
    lib.vpt_camera_default.argtypes = [C.POINTER(Camera)]
    lib.vpt_camera_default.restype = None
    lib.vpt_gpu_vdb_bounds.argtypes = [C.POINTER(GpuVdb), C.POINTER(Float3), C.POINTER(Float3)]
    lib.vpt_gpu_vdb_bounds.restype = None
    lib.vpt_instance_xform.argtypes = [C.POINTER((C.c_float * 4) * 4), C.POINTER(C.c_double * 3), C.POINTER(C.c_double * 4), C.c_double,
                                       C.POINTER((C.c_float * 4) * 4)]
    lib.vpt_instance_xform.restype = None
    lib.vpt_kernel_params_default.argtypes = [C.POINTER(KernelParams)]
    lib.vpt_kernel_params_default.restype = None
    lib.vpt_atmosphere_default_model.argtypes = [C.POINTER(AtmosphereParameters)]
    lib.vpt_atmosphere_model_options_default.argtypes = [C.POINTER(abi.AtmosphereModelOptions)]
    lib.vpt_atmosphere_model_options_default.restype = None
    lib.vpt_atmosphere_model.argtypes = [C.POINTER(abi.AtmosphereModelOptions), C.c_char_p, C.POINTER(AtmosphereParameters)]
    lib.vpt_atmosphere_precompute.argtypes = [vp, C.POINTER(AtmosphereParameters), C.c_int, vp]
    lib.vpt_atmosphere_precompute_model.argtypes = [vp, C.POINTER(abi.AtmosphereModelOptions), C.c_char_p, C.POINTER(AtmosphereParameters), C.c_int, vp]
    lib.vpt_atmosphere_read_lut.argtypes = [vp, C.POINTER(AtmosphereParameters), C.c_int, vp, C.c_size_t]
    lib.vpt_env_cdf_build.argtypes = [C.POINTER(KernelParams), C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
    lib.vpt_env_cdf_create.argtypes = [vp, C.POINTER(KernelParams)]
    # test probes (include/vpt_testhooks.h)
    lib.vpt_test_host_math.argtypes = [C.c_int, vp, vp, C.c_int]
    lib.vpt_test_device_math.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    lib.vpt_test_device_uniform_stream.argtypes = [vp, C.c_ulonglong, C.c_ulonglong, C.c_int, vp]
    lib.vpt_test_device_product_stream.argtypes = [vp, C.c_uint, C.c_uint, C.c_int, vp]
    if path is None:
        _lib = lib
    return lib


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def env_cdf_build(kp, res=180, lib=None):
    """create_cdf's tables (main.cpp:647-757) as numpy arrays: dict(val, func, cdf, marginal_func,
    marginal_cdf, marginal_int, res).  Host only."""
    lib = lib or load_library()
    val = np.zeros((res, res, 4), np.float32)
    func = np.zeros((res, res), np.float32)
    cdf = np.zeros((res, res), np.float32)
    mf = np.zeros(res, np.float32)
    mc = np.zeros(res, np.float32)
    mi = C.c_float()
    rc = lib.vpt_env_cdf_build(C.byref(kp), int(res), _np_ptr(val), _np_ptr(func), _np_ptr(cdf), _np_ptr(mf), _np_ptr(mc), C.byref(mi))
    if rc != 0:
        raise VptError("vpt_env_cdf_build -> %s" % abi.E_NAMES.get(rc, rc))
    return dict(val=val, func=func, cdf=cdf, marginal_func=mf, marginal_cdf=mc, marginal_int=float(mi.value), res=int(res))


class Context:
    """One vpt_ctx (one GPU)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.vpt_create(int(device), C.byref(h))
        if rc != 0:
            raise VptError("vpt_create(%d) -> %s: %s" % (device, abi.E_NAMES.get(rc, rc), self.lib.vpt_last_error(None).decode()))
        self.h = h
        self.device = int(device)
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.vpt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise VptError("%s -> %s: %s" % (what, abi.E_NAMES.get(rc, rc), self.lib.vpt_last_error(self.h).decode()))

    @property
    def stream(self):
        return self.lib.vpt_stream(self.h)

    def sync(self):
        self._chk(self.lib.vpt_sync(self.h), "vpt_sync")

    def texture(self, data, channels, normalized=True, linear=True, address=(abi.ADDR_CLAMP,) * 3):
        """data: numpy float32 [d, h, w(, c)] / [h, w(, c)] / [w(, c)], x fastest."""
        a = np.ascontiguousarray(data, dtype=np.float32)
        shape = a.shape[:-1] if channels == 4 else a.shape
        if channels == 4:
            assert a.shape[-1] == 4
        dims = list(shape)[::-1] + [1, 1]
        desc = TextureDesc(dims[0], dims[1], dims[2], channels, int(normalized), int(linear), (C.c_int * 3)(*address))
        out = vpt_texture_t(0)
        self._chk(self.lib.vpt_texture_create(self.h, C.byref(desc), _np_ptr(a), C.byref(out)), "vpt_texture_create")
        return out.value

    def texture_device(self, tensor, dims, channels, normalized=True, linear=True, address=(abi.ADDR_CLAMP,) * 3):
        """adopt a torch CUDA float32 tensor as a texture (no copy); dims = (w, h, d)"""
        desc = TextureDesc(dims[0], dims[1], dims[2], channels, int(normalized), int(linear), (C.c_int * 3)(*address))
        out = vpt_texture_t(0)
        self._keep.append(tensor)
        self._chk(self.lib.vpt_texture_create_device(self.h, C.byref(desc), C.c_void_p(tensor.data_ptr()), C.byref(out)), "vpt_texture_create_device")
        return out.value

    def set_volumes(self, volumes):
        arr = (GpuVdb * len(volumes))(*volumes)
        self._chk(self.lib.vpt_scene_set_volumes(self.h, arr, len(volumes)), "vpt_scene_set_volumes")

    def root(self):
        lo, hi, mx, mn = Float3(), Float3(), C.c_float(), C.c_float()
        self._chk(self.lib.vpt_scene_get_root(self.h, C.byref(lo), C.byref(hi), C.byref(mx), C.byref(mn)), "vpt_scene_get_root")
        return lo.tuple(), hi.tuple(), mx.value, mn.value

    def octree_stats(self):
        out = (C.c_int * 3)()
        self._chk(self.lib.vpt_scene_get_octree_stats(self.h, C.byref(out)), "vpt_scene_get_octree_stats")
        return list(out)

    def invalidate_sky_tables(self):
        """drop the per-view caches of the environment tail (they are rebuilt by the next render that needs them)"""
        self._chk(self.lib.vpt_invalidate_sky_tables(self.h), "vpt_invalidate_sky_tables")

    def frame_ahead_invalidate(self):
        """void the rays traced ahead of a one-iteration call sequence (a device buffer behind an unchanged pointer was rewritten in place)"""
        self._chk(self.lib.vpt_frame_ahead_invalidate(self.h), "vpt_frame_ahead_invalidate")

    def set_frame_ahead(self, on):
        self._chk(self.lib.vpt_set_frame_ahead(self.h, int(bool(on))), "vpt_set_frame_ahead")

    def set_counting(self, on):
        self._chk(self.lib.vpt_set_counting(self.h, int(bool(on))), "vpt_set_counting")

    def stats(self):
        s = RenderStats()
        self._chk(self.lib.vpt_get_stats(self.h, C.byref(s)), "vpt_get_stats")
        return s

    def render(self, cam, lights, sphere, atmosphere, kp, stream=None):
        self._chk(self.lib.vpt_render(self.h, C.byref(cam), C.byref(lights), C.byref(sphere), C.byref(atmosphere), C.byref(kp),
                                      C.c_void_p(stream) if stream else None), "vpt_render")

    def render_batch(self, cam, lights, sphere, atmosphere, kp, iter_count, iter_stride=1, stream=None):
        self._chk(self.lib.vpt_render_batch(self.h, C.byref(cam), C.byref(lights), C.byref(sphere), C.byref(atmosphere), C.byref(kp),
                                            int(iter_count), int(iter_stride), C.c_void_p(stream) if stream else None), "vpt_render_batch")

    # ---- multi-GPU: the collective lives below the C ABI (RCCL, loaded on first use) ----------------
    def comm_unique_id(self):
        """ncclGetUniqueId: 128 bytes, made on ONE rank and handed to the others by the host's own means"""
        buf = (C.c_ubyte * abi.COMM_ID_BYTES)()
        self._chk(self.lib.vpt_comm_unique_id(buf), "vpt_comm_unique_id")
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        buf = (C.c_ubyte * abi.COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._chk(self.lib.vpt_comm_init_rank(self.h, int(nranks), int(rank), buf), "vpt_comm_init_rank")
        self.comm_nranks = int(nranks)

    def comm_destroy(self):
        self._chk(self.lib.vpt_comm_destroy(self.h), "vpt_comm_destroy")
        self.comm_nranks = 0

    def allreduce_accum(self, accum_tensor, n_local, stream=None):
        """accum (running mean of this rank's n_local iterations) <- the job's mean; enqueued on `stream` (ctx stream if None)"""
        self._chk(self.lib.vpt_allreduce_accum(self.h, C.c_void_p(accum_tensor.data_ptr()), int(accum_tensor.numel()), int(n_local),
                                               C.c_void_p(stream) if stream else None), "vpt_allreduce_accum")

    def resolve_display(self, kp, stream=None):
        """display / raw.xyz of kp's buffers from its accum buffer as it is now (after an all-reduce)"""
        self._chk(self.lib.vpt_resolve_display(self.h, C.byref(kp), C.c_void_p(stream) if stream else None), "vpt_resolve_display")

    def blue_noise_advance(self, bn_tensor, steps, num_pixels, stream=None):
        """num_pixels = W*H of the render being positioned (only min(W*H, 65536) entries advance per launch)."""
        self._chk(self.lib.vpt_blue_noise_advance(self.h, C.c_void_p(bn_tensor.data_ptr()), int(steps), int(num_pixels),
                                                  C.c_void_p(stream) if stream else None), "vpt_blue_noise_advance")
