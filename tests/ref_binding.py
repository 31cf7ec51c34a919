"""ctypes binding of oracle/_ref/libvptref.so -- TEST INFRASTRUCTURE.

libvptref.so is the reference's own source/render_kernel.cu (+ source/bvh/octree.cpp) compiled for the CPU where it
lies under /root/reference, over the stand-in CUDA headers of oracle/ref_shim/ (recipe: oracle/Makefile, target ref).
It only exists where /root/reference does (this container); the GPU box gets the prebuilt file with the snapshot.
It pins the oracle: same SceneDesc, same host buffers, same oracle textures, reference code in between.
"""
import ctypes as C
import os

from oracle_binding import ROOT, OracleBinding, abi, load_oracle

REF_PATH = os.environ.get("VPT_REF_PATH", os.path.join(ROOT, "oracle", "_ref", "libvptref.so"))

_ref = None
_synthetic = {}


def have_ref():
    """True when the compiled reference is available; builds it first where the reference tree exists."""
    if not os.path.exists(REF_PATH) and os.path.isdir("/root/reference/source") and "VPT_REF_PATH" not in os.environ:
        try:
            import __graft_entry__ as ge
            ge.build_oracle()
            ge.build_ref()
        except Exception:
            return False
    return os.path.exists(REF_PATH)


def load_ref():
    global _ref
    if _ref is not None:
        return _ref
    load_oracle()                      # same liborc instance: texture handles are shared
    r = C.CDLL(REF_PATH)
    r.ref_render.argtypes = [C.POINTER(abi.Camera), C.POINTER(abi.LightList), C.POINTER(abi.GpuVdb), C.c_int, C.POINTER(abi.Sphere),
                             C.POINTER(abi.AtmosphereParameters), C.POINTER(abi.KernelParams), C.c_uint, C.c_int]
    r.ref_curand_uniform_stream.argtypes = [C.c_ulonglong, C.c_ulonglong, C.c_int, C.c_void_p]
    r.ref_curand_uniform_stream.restype = None
    _ref = r
    return r


class RefBinding(OracleBinding):
    """OracleBinding whose render() runs the compiled reference kernel instead of the restatement."""

    def __init__(self, sd):
        super().__init__(sd)
        self.r = load_ref()

    def render(self, iter_count, iter_stride=1, iteration=None, nthreads=0):
        assert iter_stride == 1
        if iteration is not None:
            self.kp.iteration = int(iteration)
        if nthreads <= 0:
            nthreads = os.cpu_count() or 1
        rc = self.r.ref_render(C.byref(self.sd.camera), C.byref(self.lights), self.volumes, len(self.volumes), C.byref(self.sd.sphere),
                               C.byref(self.atmosphere), C.byref(self.kp), int(iter_count), int(nthreads))
        if rc != 0:
            raise RuntimeError("ref_render -> %d" % rc)
        self.kp.iteration += int(iter_count)


def attach_synthetic_atmosphere(sd, seed=5):
    """Default sky scalars + smooth positive stand-in tables of the reference's sizes (CPU-only: the real tables
    come from the GPU precompute).  Good enough to pin the render path: which branches it takes depends on the
    geometry (r, mu, mu_s, nu), not on what the tables hold."""
    import numpy as np
    if seed in _synthetic:
        sd.atmosphere = pkg_atmosphere().default_model()
        sd.atm_luts = _synthetic[seed]
        return sd
    shapes = pkg_atmosphere().LUT_SHAPES
    rng = np.random.default_rng(seed)

    def smooth(shape, lo, hi):
        a = np.zeros(shape, np.float64)
        grids = np.meshgrid(*[np.linspace(0.0, 1.0, n) for n in shape[:-1]], indexing="ij")
        for c in range(4):
            acc = np.zeros(shape[:-1])
            for _ in range(3):
                k = rng.uniform(0.5, 3.0, len(grids))
                ph = rng.uniform(0.0, 6.28, len(grids))
                acc += np.prod([0.5 + 0.5 * np.sin(6.28 * kk * g + p) for kk, g, p in zip(k, grids, ph)], axis=0)
            a[..., c] = lo + (hi - lo) * acc / 3.0
        return a.astype(np.float32)

    sd.atmosphere = pkg_atmosphere().default_model()
    sd.atm_luts = {"transmittance": smooth(shapes["transmittance"], 0.05, 1.0), "irradiance": smooth(shapes["irradiance"], 0.0, 0.3),
                   "scattering": smooth(shapes["scattering"], 0.0, 0.2), "single_mie": smooth(shapes["single_mie"], 0.0, 0.1)}
    _synthetic[seed] = sd.atm_luts
    return sd


def pkg_atmosphere():
    from oracle_binding import pkg
    return pkg.atmosphere


REF_CDF_PATH = os.path.join(ROOT, "oracle", "_ref", "ref_env_cdf")


def have_ref_env_cdf():
    """the reference's create_cdf fill + host sky compiled from main.cpp's own lines (oracle/Makefile, target ref)"""
    if not os.path.exists(REF_CDF_PATH) and os.path.isdir("/root/reference/source"):
        try:
            import __graft_entry__ as ge
            ge.build_ref()
        except Exception:
            return False
    return os.path.exists(REF_CDF_PATH)


def ref_env_cdf(azimuth, elevation, sky_color):
    """run oracle/_ref/ref_env_cdf: dict(val [res,res,3], func, cdf [res,res], marginal_func, marginal_cdf [res], marginal_int, res)"""
    import subprocess
    import tempfile
    import numpy as np
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        subprocess.run([REF_CDF_PATH, repr(float(azimuth)), repr(float(elevation))] + [repr(float(c)) for c in sky_color] + [f.name], check=True)
        raw = open(f.name, "rb").read()
    res = int(np.frombuffer(raw[:4], "<u4")[0])
    mi = float(np.frombuffer(raw[4:8], "<f4")[0])
    a = np.frombuffer(raw[8:], "<f4")
    n = res * res
    return dict(res=res, marginal_int=mi, val=a[:3 * n].reshape(res, res, 3).copy(), func=a[3 * n:4 * n].reshape(res, res).copy(),
                cdf=a[4 * n:5 * n].reshape(res, res).copy(), marginal_func=a[5 * n:5 * n + res].copy(), marginal_cdf=a[5 * n + res:5 * n + 2 * res].copy())
