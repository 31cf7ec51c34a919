"""vpt_atmosphere_model (csrc/vpt_atmosphere.hip: the sky-model scalars for any setting of the reference's switches) against
the reference's own model set-up compiled from atmosphere.cpp's lines (oracle/_ref/ref_atmosphere_model, see
oracle/ref_shim/ref_model_driver.cpp): every scalar member of AtmosphereParameters bit for bit -- live where the program
exists, and against tests/golden/ref_atmosphere_model.npz (written from it by this file's __main__) everywhere."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROG = os.path.join(ROOT, "oracle", "_ref", "ref_atmosphere_model")
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_atmosphere_model.npz")

# (constant solar, ozone, white balance, use_luminance, exposure, lambdas)
CASES = {
    "defaults": (1, 1, 1, 0, 1.0, (680.0, 550.0, 440.0)),
    "astm_solar_no_ozone": (0, 0, 1, 0, 1.0, (680.0, 550.0, 440.0)),
    "off_node_wavelengths": (0, 1, 0, 0, 2.5, (612.3, 549.1, 465.7)),
    "approximate_luminance": (0, 1, 1, 1, 10.0, (650.0, 510.0, 475.0)),
    "edge_wavelengths": (1, 1, 1, 0, 1.0, (830.0, 360.0, 355.0)),
    "precomputed_luminance": (1, 1, 1, 2, 1.0, (680.0, 550.0, 440.0)),
    "precomputed_luminance_pass_2": (0, 1, 1, 2, 1.0, (563.666666, 595.0, 626.333333)),      # a wavelength triple of the five-pass precompute
}


def have_prog():
    if not os.path.exists(PROG) and os.path.isdir("/root/reference/source"):
        try:
            import __graft_entry__ as ge
            ge.build_ref()
        except Exception:
            return False
    return os.path.exists(PROG)


def run_reference(case):
    cs, oz, wb, lum, ex, lam = case
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        subprocess.run([PROG, str(cs), str(oz), str(wb), str(lum), repr(ex)] + [repr(x) for x in lam] + [f.name], check=True)
        return np.frombuffer(open(f.name, "rb").read(), np.uint8).copy()


def product_bytes(pkg, case):
    """the same members, in the order ref_model_driver.cpp writes them"""
    cs, oz, wb, lum, ex, lam = case
    p = pkg.atmosphere.model(use_constant_solar_spectrum=cs, use_ozone=oz, do_white_balance=wb, use_luminance=lum, exposure=ex, lambdas=lam)
    out = bytearray()

    def f3(v):
        out.extend(np.array([v.x, v.y, v.z], np.float32).tobytes())

    def prof(d):
        for i in range(2):
            l = d.layers[i]
            out.extend(np.array([l.width, l.exp_term, l.exp_scale, l.linear_term, l.const_term], np.float32).tobytes())
    f3(p.sky_spectral_radiance_to_luminance); f3(p.sun_spectral_radiance_to_luminance); f3(p.solar_irradiance)
    out.extend(np.array([p.sun_angular_radius, p.bottom_radius, p.top_radius], np.float32).tobytes())
    prof(p.rayleigh_density); f3(p.rayleigh_scattering)
    prof(p.mie_density); f3(p.mie_scattering); f3(p.mie_extinction)
    out.extend(np.float32(p.mie_phase_function_g).tobytes())
    prof(p.absorption_density); f3(p.absorption_extinction); f3(p.ground_albedo)
    out.extend(np.float32(p.mu_s_min).tobytes())
    out.extend(np.int32(p.use_luminance).tobytes())
    f3(p.white_point)
    out.extend(np.float32(p.exposure).tobytes())
    return np.frombuffer(bytes(out), np.uint8)


def _same(a, b, name):
    assert a.size == b.size, name
    if not np.array_equal(a, b):
        fa, fb = a.view(np.float32), b.view(np.float32)
        bad = np.flatnonzero(fa.view(np.uint32) != fb.view(np.uint32))
        raise AssertionError("%s: words %s differ: %s vs %s" % (name, bad.tolist(), fa[bad].tolist(), fb[bad].tolist()))


def test_model_matches_reference_golden(pkg):
    g = np.load(GOLDEN)
    assert sorted(g.files) == sorted(CASES)
    for name, case in CASES.items():
        _same(product_bytes(pkg, case), g[name], name)


def test_model_matches_reference_live(pkg):
    if not have_prog():
        pytest.skip("oracle/_ref/ref_atmosphere_model not available (no reference tree here)")
    for name, case in CASES.items():
        _same(product_bytes(pkg, case), run_reference(case), name)
    _same(product_bytes(pkg, (0, 1, 1, 0, 0.25, (701.5, 523.25, 401.0))), run_reference((0, 1, 1, 0, 0.25, (701.5, 523.25, 401.0))), "extra")


def test_default_options_reproduce_the_builtin_default_model(pkg):
    assert bytes(pkg.atmosphere.model()) == bytes(pkg.atmosphere.default_model())
    lum = pkg.atmosphere.model(use_luminance=2)                   # PRECOMPUTED: the sky factor is MAX_LUMINOUS_EFFICACY alone
    assert lum.use_luminance == 2 and lum.sky_spectral_radiance_to_luminance.x == 683.0 == lum.sky_spectral_radiance_to_luminance.z
    with pytest.raises(pkg.VptError):
        pkg.atmosphere.model(spectra_file="/nonexistent/spectra.bin")


if __name__ == "__main__":
    assert have_prog()
    np.savez_compressed(GOLDEN, **{name: run_reference(case) for name, case in CASES.items()})
    print("wrote", GOLDEN, os.path.getsize(GOLDEN), "bytes")
