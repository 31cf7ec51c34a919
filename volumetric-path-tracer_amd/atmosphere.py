"""Host side of the procedural-sky prerequisite: default model + LUT precompute through the C ABI
(reference source/atmosphere/atmosphere.cpp:1177 `init`).  The tables are INPUTS of the hot path:
they are generated once on the GPU and handed to both the HIP renderer and (as numpy arrays) the
CPU oracle."""
import ctypes as C

import numpy as np

from .abi import AtmosphereParameters
from .host import Context, VptError, load_library

LUT_SHAPES = {"transmittance": (64, 256, 4), "irradiance": (64, 256, 4), "scattering": (32, 128, 256, 4), "single_mie": (32, 128, 256, 4)}
_WHICH = {"transmittance": 0, "irradiance": 1, "scattering": 2, "single_mie": 3}
_cache = {}


def default_model():
    p = AtmosphereParameters()
    rc = load_library().vpt_atmosphere_default_model(C.byref(p))
    if rc != 0:
        raise VptError("vpt_atmosphere_default_model -> %d" % rc)
    return p


def model(spectra_file=None, **options):
    """vpt_atmosphere_model: the scalars for any switch setting (use_constant_solar_spectrum, use_ozone, do_white_balance,
    use_luminance, half_precision, exposure, lambdas=(r, g, b) in nm, length_unit_in_meters)."""
    from . import abi
    lib = load_library()
    o = abi.AtmosphereModelOptions()
    lib.vpt_atmosphere_model_options_default(C.byref(o))
    for k, v in options.items():
        if k == "lambdas":
            o.lambdas = (C.c_double * 3)(*[float(x) for x in v])
        else:
            setattr(o, k, v)
    p = AtmosphereParameters()
    rc = lib.vpt_atmosphere_model(C.byref(o), spectra_file.encode() if spectra_file else None, C.byref(p))
    if rc != 0:
        raise VptError("vpt_atmosphere_model -> %d" % rc)
    return p


def model_options(**options):
    from . import abi
    o = abi.AtmosphereModelOptions()
    load_library().vpt_atmosphere_model_options_default(C.byref(o))
    for k, v in options.items():
        if k == "lambdas":
            o.lambdas = (C.c_double * 3)(*[float(x) for x in v])
        else:
            setattr(o, k, v)
    return o


def _read_luts(ctx, p):
    luts = {}
    for name, shape in LUT_SHAPES.items():
        a = np.empty(shape, np.float32)
        ctx._chk(ctx.lib.vpt_atmosphere_read_lut(ctx.h, C.byref(p), _WHICH[name], a.ctypes.data_as(C.c_void_p), a.size), "vpt_atmosphere_read_lut")
        luts[name] = a
    return luts


def precompute_model(ctx, orders=4, spectra_file=None, **options):
    """vpt_atmosphere_precompute_model: model + table passes for any luminance mode (use_luminance=2: the PRECOMPUTED mode's five
    passes over 15 wavelengths); returns (params, dict of numpy LUTs)."""
    p = AtmosphereParameters()
    o = model_options(**options)
    ctx._chk(ctx.lib.vpt_atmosphere_precompute_model(ctx.h, C.byref(o), spectra_file.encode() if spectra_file else None, C.byref(p), int(orders), None),
             "vpt_atmosphere_precompute_model")
    return p, _read_luts(ctx, p)


def precompute(ctx, params=None, orders=4):
    """Runs the precompute on ctx's GPU; returns (params with device buffers + textures, dict of numpy LUTs)."""
    p = params or default_model()
    ctx._chk(ctx.lib.vpt_atmosphere_precompute(ctx.h, C.byref(p), int(orders), None), "vpt_atmosphere_precompute")
    luts = {}
    for name, shape in LUT_SHAPES.items():
        a = np.empty(shape, np.float32)
        ctx._chk(ctx.lib.vpt_atmosphere_read_lut(ctx.h, C.byref(p), _WHICH[name], a.ctypes.data_as(C.c_void_p), a.size), "vpt_atmosphere_read_lut")
        luts[name] = a
    return p, luts


def attach_default_atmosphere(sd, device=0, **model_options):
    """Fill sd.atmosphere (scalars) and sd.atm_luts (numpy tables) with the reference's default sky -- or, with model_options, the
    sky of model(**model_options); the tables are computed once per process, device and option set."""
    key = (int(device), tuple(sorted((k, tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in model_options.items())))
    if key not in _cache:
        ctx = Context(device)
        if model_options.get("use_luminance") == 2:
            p, luts = precompute_model(ctx, **model_options)
        else:
            p, luts = precompute(ctx, model(**model_options) if model_options else None)
        scal = AtmosphereParameters.from_buffer_copy(p)
        for f in ("delta_irradience_buffer", "delta_rayleigh_scattering_buffer", "delta_mie_scattering_buffer", "delta_scattering_density_buffer",
                  "delta_multiple_scattering_buffer", "transmittance_buffer", "irradiance_buffer", "scattering_buffer", "optional_mie_single_scattering_buffer"):
            setattr(scal, f, None)
        for f in ("transmittance_texture", "scattering_texture", "irradiance_texture", "single_mie_scattering_texture"):
            setattr(scal, f, 0)
        _cache[key] = (scal, luts)
        ctx.close()       # frees nothing the tables need: they were read back to the host
    scal, luts = _cache[key]
    sd.atmosphere = AtmosphereParameters.from_buffer_copy(scal)
    sd.atm_luts = luts
    return sd
