"""Scene assembly above the C ABI: the host-side steps the reference performs between
loading a grid and launching the kernel, restated so that tests / bench / smoke build the
exact POD inputs `volume_rt_kernel` receives.

  make_vdb_info       GPU_VDB::loadVDB's info block      (gpu_vdb.cpp:171-212, 413-471)
  frame_camera        the "F key" framing                (main.cpp:525-543)
  default_sphere      the reference sphere               (main.cpp:1480-1488)
  SceneDesc           backend-neutral bundle (numpy + PODs) consumed by HipBinding
                      (this file) and by tests/oracle_binding.py
"""
import ctypes as C
import math
import os

import numpy as np

from . import abi
from .abi import (AtmosphereParameters, Camera, Float3, GpuVdb, Int3, KernelParams, LightList, PointLight, Sphere)
from .host import Context, load_library

FLT_EPSILON = np.float32(1.1920929e-07)
FLT_MAX = np.float32(3.4028235e38)

# scene assets of the package itself: decoded copies of the reference's dragon.vdb grid, BN0.bmp and its two look-up EXRs (written by
# tests/golden/make_fixtures.py), so that the product never reads from the test tree; the test-only fixtures stay in tests/golden
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def f3(v):
    return Float3(float(v[0]), float(v[1]), float(v[2]))


def make_gpu_vdb(density, bbox_min, bbox_max, matrix, voxel_size, emission=None, color=None):
    """VDB_INFO + xform exactly as GPU_VDB::loadVDB fills them (texture handles left 0).

    density: float32 [z, y, x] dense copy over the active bbox.  matrix: OpenVDB Mat4d
    (row-vector convention).  xform[i][j] = matrix(j, i) (gpu_vdb.cpp:81-92).
    """
    v = GpuVdb()
    vi = v.vdb_info
    if isinstance(density, np.ndarray):
        d = np.ascontiguousarray(density, dtype=np.float32)
        dmax, dmin = d.max(), np.maximum(FLT_EPSILON, d).min()
    else:                                                                       # torch tensor already in HBM
        d = density
        dmax, dmin = np.float32(d.max().item()), np.float32(d.clamp(min=float(FLT_EPSILON)).min().item())
    vi.voxelsize = float(np.float32(voxel_size))
    vi.dim = Int3(int(d.shape[2]), int(d.shape[1]), int(d.shape[0]))
    vi.bmin = f3(np.asarray(bbox_min, dtype=np.float32))
    vi.bmax = f3(np.asarray(bbox_max, dtype=np.float32))
    vi.max_density = float(max(np.float32(0.0), dmax))                          # gpu_vdb.cpp:206
    vi.min_density = float(min(FLT_MAX, dmin))                                  # gpu_vdb.cpp:207
    vi.has_color = 1 if color is not None else 0
    vi.has_emission = 1 if emission is not None else 0
    m = np.asarray(matrix, dtype=np.float64).astype(np.float32)
    for i in range(4):
        for j in range(4):
            v.xform[i][j] = float(m[j, i])
    return v


def frame_camera(lib, volumes, width, height, fov=30.0, aperture=0.0):
    """main.cpp:525-543: bbox (seeded with the origin!) of the transformed bmin/bmax corners,
    lookat = centre, lookfrom = centre + |diag| * (1,1,1)."""
    bmin = np.zeros(3, np.float32)
    bmax = np.zeros(3, np.float32)
    for v in volumes:
        xf = np.array([[v.xform[c][r] for c in range(4)] for r in range(4)], dtype=np.float32)   # M[r][c] = m[c][r]
        xt = xf.T                                                                               # xform.transpose()
        def tp(p):
            p4 = np.array([p.x, p.y, p.z, 1.0], np.float32)
            return (xt @ p4)[:3].astype(np.float32)
        bmin = np.minimum(bmin, tp(v.vdb_info.bmin))
        bmax = np.maximum(bmax, tp(v.vdb_info.bmax))
    center = ((bmax + bmin) / np.float32(2)).astype(np.float32)
    dist = np.float32(np.sqrt(np.sum((bmax - bmin).astype(np.float32) ** 2, dtype=np.float32)))
    lookfrom = (center + dist).astype(np.float32)
    cam = Camera()
    lib.vpt_camera_default(C.byref(cam))
    lib.vpt_camera_update(C.byref(cam), f3(lookfrom), f3(center), Float3(0, 1, 0), float(fov), float(width) / float(height), float(aperture))
    return cam, center, dist


def default_sphere():
    s = Sphere()                                  # main.cpp:1480-1488
    s.center = Float3(0, 1000, 0)
    s.radius = 1.0
    s.color = Float3(10.0, 0, 0)
    s.roughness = 1.0
    return s


def load_golden(name):
    p = os.path.join(DATA_DIR, name)
    return np.load(p if os.path.exists(p) else os.path.join(GOLDEN_DIR, name))


def blue_noise_from_rgb(rgb):
    """load_texture_bmp_gpu (fileIO.cpp:460-495): x = R/255, y = B/255, z = G/255."""
    rgb = rgb.astype(np.float32)
    out = np.empty((256 * 256, 3), np.float32)
    flat = rgb.reshape(-1, 3)
    out[:, 0] = flat[:, 0] / np.float32(255.0)
    out[:, 1] = flat[:, 2] / np.float32(255.0)
    out[:, 2] = flat[:, 1] / np.float32(255.0)
    return out


class SceneDesc:
    """Backend-neutral scene: numpy grids + PODs without texture handles / buffer pointers."""

    def __init__(self):
        self.volumes = []          # list of (GpuVdb, density[z,y,x], emission or None, color[z,y,x,4] or None)
        self.camera = None
        self.lights = []           # list of PointLight
        self.sphere = default_sphere()
        self.kp = KernelParams()
        self.atmosphere = AtmosphereParameters()
        self.atm_luts = None       # dict transmittance/irradiance/scattering/single_mie numpy float4 arrays
        self.env_map = None        # numpy [h, w, 4]
        self.env_cdf = None        # host.env_cdf_build() tables (integrator != 0 on the procedural sky)
        self.blue_noise = None     # numpy [65536, 3]
        self.emission_lut = None   # numpy [256, 3]
        self.density_color_lut = None
        self.width = 0
        self.height = 0


def dragon_scene(width, height, config="c1", lib=None):
    """The BASELINE.md configs on assets/dragon.vdb (via the committed golden fixture).

    c1: one point light above the bbox, sun_mult = sky_mult = 0        
    c2: procedural sun + sky (needs atmosphere LUTs bound by the caller) (row 2)
    sun: c2 without the sky term (sun NEE only, sky_mult = 0) -- no LUTs needed
    """
    lib = lib or load_library()
    g = load_golden("dragon_dense.npz")
    luts = load_golden("luts.npz")
    bn = load_golden("bn0.npz")
    sd = SceneDesc()
    sd.width, sd.height = int(width), int(height)
    vdb = make_gpu_vdb(g["density"], g["bbox_min"], g["bbox_max"], g["matrix"], g["voxel_size"])
    sd.volumes.append((vdb, np.ascontiguousarray(g["density"], np.float32), None, None))
    cam, center, dist = frame_camera(lib, [vdb], width, height)
    sd.camera = cam
    kp = KernelParams()
    lib.vpt_kernel_params_default(C.byref(kp))
    kp.resolution = abi.UInt2(int(width), int(height))
    kp.max_interactions = 1 << 30
    if config == "c1":
        kp.sun_mult = 0.0
        kp.sky_mult = 0.0
        pl = PointLight()
        pl.pos = f3(center + np.array([0, dist, 0], np.float32))
        pl.color = Float3(1, 1, 1)
        pl.power = float(np.float32(dist) * np.float32(dist))
        sd.lights.append(pl)
    elif config == "sun":
        kp.sky_mult = 0.0
    elif config == "c2":
        pass
    else:
        raise ValueError(config)
    sd.kp = kp
    sd.blue_noise = blue_noise_from_rgb(bn["rgb"])
    sd.emission_lut = np.ascontiguousarray(luts["blackbody"], np.float32)
    sd.density_color_lut = np.ascontiguousarray(luts["density_color"], np.float32)
    return sd


def bind_env_cdf(kp, t, make_texture):
    """the five sampler states of create_cdf (main.cpp:775-867) on either backend"""
    wc = (abi.ADDR_WRAP, abi.ADDR_CLAMP, abi.ADDR_WRAP)
    kp.sky_tex = make_texture(t["val"], 4, address=wc)
    kp.env_func_tex = make_texture(t["func"], 1, normalized=False, linear=False, address=wc)
    kp.env_cdf_tex = make_texture(t["cdf"], 1, normalized=False, linear=False, address=wc)
    kp.env_marginal_func_tex = make_texture(t["marginal_func"], 1, normalized=False, linear=False, address=wc)
    kp.env_marginal_cdf_tex = make_texture(t["marginal_cdf"], 1, normalized=False, linear=False, address=wc)
    kp.env_sample_tex_res = int(t["res"])
    kp.env_marginal_int = float(t["marginal_int"])


class HipBinding:
    """Uploads a SceneDesc through the C ABI and owns the torch HBM buffers the reference's
    main loop would own (accum/raw/cost/depth/display, main.cpp:596-637)."""

    def __init__(self, sd, device=0, ctx=None):
        import torch
        self.torch = torch
        self.sd = sd
        self.ctx = ctx or Context(device)
        self.dev = torch.device("cuda", self.ctx.device)
        ctx = self.ctx
        vols = []
        shared = {}                          # instances of one file share its textures (main.cpp:1064)
        for vdb, dens, emis, col in sd.volumes:
            v = GpuVdb.from_buffer_copy(vdb)
            if isinstance(dens, np.ndarray):
                tex_key = id(dens)
                if tex_key not in shared:
                    shared[tex_key] = ctx.texture(dens, 1)
                v.vdb_info.density_texture = shared[tex_key]
            else:                            # torch tensor in HBM: adopted without a copy
                v.vdb_info.density_texture = ctx.texture_device(dens, (dens.shape[2], dens.shape[1], dens.shape[0]), 1)
            if emis is not None:
                v.vdb_info.emission_texture = ctx.texture(emis, 1)
            if col is not None:
                if id(col) not in shared:
                    shared[id(col)] = ctx.texture(col, 4)
                v.vdb_info.color_texture = shared[id(col)]
            vols.append(v)
        self.volumes = vols
        ctx.set_volumes(vols)
        n = sd.width * sd.height
        t = torch
        self.accum = t.zeros(n, 3, dtype=t.float32, device=self.dev)
        self.cost = t.zeros(n, 3, dtype=t.float32, device=self.dev)
        self.depth = t.zeros(n, dtype=t.float32, device=self.dev)
        self.raw = t.zeros(n, 4, dtype=t.float32, device=self.dev)
        self.display = t.zeros(n, dtype=t.int32, device=self.dev)
        self.blue_noise = t.from_numpy(sd.blue_noise.copy()).to(self.dev)
        self.emission_lut = t.from_numpy(sd.emission_lut.copy()).to(self.dev)
        self.density_color_lut = t.from_numpy(sd.density_color_lut.copy()).to(self.dev)
        kp = KernelParams.from_buffer_copy(sd.kp)
        kp.accum_buffer = self.accum.data_ptr()
        kp.cost_buffer = self.cost.data_ptr()
        kp.depth_buffer = self.depth.data_ptr()
        kp.raw_buffer = self.raw.data_ptr()
        kp.display_buffer = self.display.data_ptr()
        kp.blue_noise_buffer = self.blue_noise.data_ptr()
        kp.emission_texture = self.emission_lut.data_ptr()
        kp.density_color_texture = self.density_color_lut.data_ptr()
        self.kp = kp
        self.atmosphere = AtmosphereParameters.from_buffer_copy(sd.atmosphere)
        if sd.atm_luts is not None:
            L = sd.atm_luts
            wc = (abi.ADDR_WRAP, abi.ADDR_CLAMP, abi.ADDR_CLAMP)
            self.atmosphere.transmittance_texture = ctx.texture(L["transmittance"], 4, address=wc)
            self.atmosphere.irradiance_texture = ctx.texture(L["irradiance"], 4, address=wc)
            self.atmosphere.scattering_texture = ctx.texture(L["scattering"], 4)
            self.atmosphere.single_mie_scattering_texture = ctx.texture(L["single_mie"], 4)
        if sd.env_map is not None:
            kp.env_tex = ctx.texture(sd.env_map, 4, address=(abi.ADDR_WRAP, abi.ADDR_CLAMP, abi.ADDR_CLAMP))
        if sd.env_cdf is not None:
            bind_env_cdf(kp, sd.env_cdf, ctx.texture)
        self._lights_arr = (PointLight * max(1, len(sd.lights)))(*sd.lights)
        self.lights = LightList(len(sd.lights), C.cast(self._lights_arr, C.POINTER(PointLight)))
        # buffers were filled on torch's stream; the ctx renders on its own stream
        torch.cuda.synchronize(self.dev)

    def render(self, iter_count, iter_stride=1, iteration=None, stream=None):
        if iteration is not None:
            self.kp.iteration = int(iteration)
        self.ctx.render_batch(self.sd.camera, self.lights, self.sd.sphere, self.atmosphere, self.kp, iter_count, iter_stride, stream)
        self.kp.iteration += int(iter_count) * int(iter_stride)

    def render_frame(self, stream=None):
        """the reference's per-frame call (main.cpp:1822-1829): ONE iteration through vpt_render"""
        self.ctx.render(self.sd.camera, self.lights, self.sd.sphere, self.atmosphere, self.kp, stream)
        self.kp.iteration += 1

    def sync(self):
        self.ctx.sync()


# ---- synthetic stand-ins for the assets the reference does not ship -----------------
def value_noise(shape, cells, seed, octaves=1):
    """Deterministic trilinear value noise in [0, 1] on a [z, y, x] grid: a random lattice of
    `cells` periods per axis (doubling per octave, amplitude halving)."""
    rng = np.random.default_rng(seed)
    out = np.zeros(shape, np.float32)
    amp, total = 1.0, 0.0
    for o in range(octaves):
        c = cells * (1 << o)
        lat = rng.random((c + 1, c + 1, c + 1), dtype=np.float32)
        ax = [np.linspace(0, c, n, endpoint=False, dtype=np.float32) for n in shape]
        i = [np.minimum(a.astype(np.int32), c - 1) for a in ax]
        f = [(a - ii).astype(np.float32) for a, ii in zip(ax, i)]
        f = [t * t * (3 - 2 * t) for t in f]
        iz, iy, ix = np.ix_(i[0], i[1], i[2])
        fz, fy, fx = np.ix_(f[0], f[1], f[2])
        v = ((lat[iz, iy, ix] * (1 - fx) + lat[iz, iy, ix + 1] * fx) * (1 - fy) +
             (lat[iz, iy + 1, ix] * (1 - fx) + lat[iz, iy + 1, ix + 1] * fx) * fy) * (1 - fz) + \
            ((lat[iz + 1, iy, ix] * (1 - fx) + lat[iz + 1, iy, ix + 1] * fx) * (1 - fy) +
             (lat[iz + 1, iy + 1, ix] * (1 - fx) + lat[iz + 1, iy + 1, ix + 1] * fx) * fy) * fz
        out += np.float32(amp) * v.astype(np.float32)
        total += amp
        amp *= 0.5
    return (out / np.float32(total)).astype(np.float32)


def _radial(shape):
    z, y, x = [np.linspace(-1, 1, n, dtype=np.float32) for n in shape]
    return np.sqrt(z[:, None, None] ** 2 + y[None, :, None] ** 2 + x[None, None, :] ** 2).astype(np.float32)


def _grid_matrix(n, voxel, centre=(0.0, 0.0, 0.0)):
    """OpenVDB-style index->world Mat4d (row-vector convention): uniform scale + translation
    so that the grid is centred at `centre`."""
    m = np.eye(4)
    m[0, 0] = m[1, 1] = m[2, 2] = voxel
    m[3, :3] = np.asarray(centre, np.float64) - 0.5 * voxel * (np.asarray(n[::-1], np.float64) - 1)
    return m


def fireball_grids(n=256, seed=1234):
    """BASELINE config 3 stand-in for the missing fireball.vdb: density = smooth radial
    falloff x value noise, heat = density^2."""
    shape = (n, n, n)
    r = _radial(shape)
    fall = np.clip(1.0 - r, 0.0, 1.0).astype(np.float32)
    fall = fall * fall * (3 - 2 * fall)
    dens = (fall * value_noise(shape, 4, seed, octaves=3) * np.float32(2.0)).astype(np.float32)
    dens[dens < 0.02] = 0.0
    return dens, (dens * dens).astype(np.float32)


def smoke_grids(n=128, seed=4321):
    """BASELINE config 5 stand-in for colored_smoke.vdb: density + Cd (vec3 -> float4, w = 1,
    gpu_vdb.cpp:360-370)."""
    shape = (n, n, n)
    r = _radial(shape)
    fall = np.clip(1.15 - r, 0.0, 1.0).astype(np.float32)
    dens = (fall * value_noise(shape, 3, seed, octaves=2) * np.float32(1.5)).astype(np.float32)
    dens[dens < 0.05] = 0.0
    cd = np.empty(shape + (4,), np.float32)
    cd[..., 0] = value_noise(shape, 2, seed + 1)
    cd[..., 1] = value_noise(shape, 2, seed + 2)
    cd[..., 2] = value_noise(shape, 2, seed + 3)
    cd[..., 3] = 1.0
    return dens, cd


def cloud_grid(shape=(1216, 704, 1024), seed=42, occupancy=0.35, chunk=64):
    """BASELINE config 4 stand-in for the Disney cloud: fBm value noise thresholded to
    ~`occupancy`.  shape is [z, y, x]."""
    d = value_noise(shape, 4, seed, octaves=4)
    thr = np.quantile(d[::4, ::4, ::4], 1.0 - occupancy)
    d = np.maximum(d - np.float32(thr), 0.0).astype(np.float32)
    d *= np.float32(1.0) / max(np.float32(1e-6), d.max())
    return d


def cloud_grid_torch(shape=(1216, 704, 1024), seed=42, occupancy=0.35, device="cuda", chunk=64):
    """cloud_grid evaluated on the GPU with torch (bench-only generator for the full-size config 4
    grid, 3.5 GB): same construction -- 4 octaves of trilinear value noise on random lattices,
    thresholded to ~`occupancy`, normalised to max 1 -- returned as a CUDA float32 tensor [z, y, x]."""
    import torch
    rng = np.random.default_rng(seed)
    nz, ny, nx = shape
    out = torch.empty(shape, dtype=torch.float32, device=device)
    lats = []
    for o in range(4):
        c = 4 * (1 << o)
        lats.append(torch.from_numpy(rng.random((c + 1, c + 1, c + 1), dtype=np.float32)).to(device))

    def axis(n, c):
        a = torch.arange(n, device=device, dtype=torch.float32) * (c / n)
        i = torch.clamp(a.to(torch.int64), max=c - 1)
        f = a - i.to(torch.float32)
        return i, f * f * (3 - 2 * f)

    for z0 in range(0, nz, chunk):
        z1 = min(nz, z0 + chunk)
        acc = torch.zeros((z1 - z0, ny, nx), dtype=torch.float32, device=device)
        amp, total = 1.0, 0.0
        for o, lat in enumerate(lats):
            c = lat.shape[0] - 1
            iz, fz = axis(nz, c); iy, fy = axis(ny, c); ix, fx = axis(nx, c)
            iz, fz = iz[z0:z1], fz[z0:z1]
            # separable trilinear: interpolate x, then y, then z
            lx = lat[:, :, ix] * (1 - fx) + lat[:, :, ix + 1] * fx                      # [c+1, c+1, nx]
            ly = lx[:, iy, :] * (1 - fy)[None, :, None] + lx[:, iy + 1, :] * fy[None, :, None]     # [c+1, ny, nx]
            v = ly[iz] * (1 - fz)[:, None, None] + ly[iz + 1] * fz[:, None, None]
            acc += amp * v
            total += amp
            amp *= 0.5
        out[z0:z1] = acc / total
    sub = out[::4, ::4, ::4].flatten()
    thr = torch.quantile(sub[torch.randperm(sub.numel(), device=device)[:4_000_000]], 1.0 - occupancy)
    out.sub_(thr).clamp_(min=0.0)
    out.mul_(1.0 / max(1e-6, float(out.max())))
    return out


def hdri_map(w=2048, h=1024, seed=7):
    """Synthetic lat-long HDRI: sky gradient + ground + a sun lobe."""
    rng = np.random.default_rng(seed)
    v = (np.arange(h, dtype=np.float32) + 0.5) / h
    u = (np.arange(w, dtype=np.float32) + 0.5) / w
    theta = v[:, None] * np.float32(np.pi)
    phi = (u[None, :] - 0.5) * np.float32(2 * np.pi)
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) * np.ones_like(phi), np.sin(theta) * np.sin(phi)], -1).astype(np.float32)
    sun = np.array([0.5, 0.6, 0.62], np.float32)
    sun /= np.linalg.norm(sun)
    c = np.clip((d * sun).sum(-1), -1, 1)
    sky = np.where(d[..., 1:2] > 0, np.array([0.35, 0.55, 0.95], np.float32) * (0.4 + 0.6 * d[..., 1:2]),
                   np.array([0.25, 0.22, 0.2], np.float32) * (0.6 + 0.4 * d[..., 1:2]))
    lobe = (np.exp((c - 1.0) * 400.0) * 60.0 + np.exp((c - 1.0) * 8.0) * 0.8)[..., None] * np.array([1.0, 0.92, 0.8], np.float32)
    img = np.empty((h, w, 4), np.float32)
    img[..., :3] = sky + lobe + rng.random((h, w, 1), dtype=np.float32) * 0.01
    img[..., 3] = 1.0
    return img


def _base_kp(lib, width, height):
    kp = KernelParams()
    lib.vpt_kernel_params_default(C.byref(kp))
    kp.resolution = abi.UInt2(int(width), int(height))
    kp.max_interactions = 1 << 30
    return kp


def _finish(sd):
    luts = load_golden("luts.npz")
    bn = load_golden("bn0.npz")
    sd.blue_noise = blue_noise_from_rgb(bn["rgb"])
    sd.emission_lut = np.ascontiguousarray(luts["blackbody"], np.float32)
    sd.density_color_lut = np.ascontiguousarray(luts["density_color"], np.float32)
    return sd


def fireball_scene(width, height, n=256, lib=None, sky=False):
    """BASELINE config 3: emission + blackbody LUT (emission_scale = 1, pivot = 1).  sky=True is the BASELINE.md 4
    specification (procedural sun + sky: the caller binds the atmosphere LUTs, as for config 2); sky=False is the
    sun-only variant (sky_mult = 0, no LUTs needed) the small parity scenes use."""
    lib = lib or load_library()
    dens, heat = fireball_grids(n)
    sd = SceneDesc()
    sd.width, sd.height = int(width), int(height)
    voxel = 20.0 / n
    vdb = make_gpu_vdb(dens, (0, 0, 0), (n - 1, n - 1, n - 1), _grid_matrix(dens.shape, voxel), voxel, emission=heat)
    sd.volumes.append((vdb, dens, heat, None))
    sd.camera, center, dist = frame_camera(lib, [vdb], width, height)
    kp = _base_kp(lib, width, height)
    kp.emission_scale = 1.0
    kp.emission_pivot = 1.0
    if not sky:
        kp.sky_mult = 0.0
    sd.kp = kp
    return _finish(sd)


def instanced_scene(width, height, n=128, grid=10, seed=99, aperture=2.0, spacing=None, lib=None, sky=False, rotate=True, grids=None):
    """BASELINE config 5: grid x grid instances of one coloured-smoke grid (density + Cd) on a
    jittered lattice with random unit quaternions, scale 1, via the .ins transform
    (vpt_instance_xform == main.cpp:1060-1095), DOF on.  sky=True: procedural sun + sky as BASELINE.md 4 specifies
    (the caller binds the atmosphere LUTs); sky=False: sun only.  rotate=False keeps the instances axis-aligned
    (identity quaternion), the case in which an instance's AABB is exactly its index-space box.  grids: (density, Cd)
    to use instead of smoke_grids(n)."""
    lib = lib or load_library()
    dens, cd = grids if grids is not None else smoke_grids(n)
    voxel = 8.0 / n
    base = make_gpu_vdb(dens, (0, 0, 0), (n - 1, n - 1, n - 1), _grid_matrix(dens.shape, voxel), voxel, color=cd)
    rng = np.random.default_rng(seed)
    spacing = spacing or 9.0
    sd = SceneDesc()
    sd.width, sd.height = int(width), int(height)
    F44 = (C.c_float * 4) * 4
    for gi in range(grid):
        for gj in range(grid):
            pos = np.array([(gi - (grid - 1) / 2) * spacing, 0.0, (gj - (grid - 1) / 2) * spacing]) + rng.uniform(-2.0, 2.0, 3)
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            if not rotate:
                q = np.array([0.0, 0.0, 0.0, 1.0])
            v = GpuVdb.from_buffer_copy(base)
            out = F44()
            lib.vpt_instance_xform(C.byref(base.xform), C.byref((C.c_double * 3)(*pos)), C.byref((C.c_double * 4)(*q)), 1.0, C.byref(out))
            C.memmove(C.byref(v.xform), C.byref(out), C.sizeof(out))
            sd.volumes.append((v, dens, None, cd))
    vols = [v for v, _, _, _ in sd.volumes]
    sd.camera, center, dist = frame_camera(lib, vols, width, height, aperture=aperture)
    kp = _base_kp(lib, width, height)
    if not sky:
        kp.sky_mult = 0.0
    sd.kp = kp
    return _finish(sd)


def cloud_scene(width, height, shape=(152, 88, 128), env=(512, 256), integrator=1, lib=None, device_grid=None):
    """BASELINE config 4: large fBm cloud, synthetic lat-long HDRI (environment_type = 1),
    vol_integrator.  The caller binds atmosphere LUTs (vol_integrator's tail is always the
    procedural sky, render_kernel.cu:1752)."""
    lib = lib or load_library()
    dens = device_grid if device_grid is not None else cloud_grid(shape)
    shape = tuple(int(x) for x in dens.shape)
    sd = SceneDesc()
    sd.width, sd.height = int(width), int(height)
    voxel = 40.0 / shape[2]
    nz, ny, nx = shape
    vdb = make_gpu_vdb(dens, (0, 0, 0), (nx - 1, ny - 1, nz - 1), _grid_matrix(dens.shape, voxel), voxel)
    sd.volumes.append((vdb, dens, None, None))
    sd.camera, center, dist = frame_camera(lib, [vdb], width, height)
    kp = _base_kp(lib, width, height)
    kp.environment_type = 1
    kp.integrator = int(integrator)
    sd.kp = kp
    sd.env_map = hdri_map(*env)
    return _finish(sd)
