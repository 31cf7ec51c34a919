"""vpt_env_cdf_build (host code, csrc/vpt_env.hip) against an independent numpy restatement of
create_cdf + the host single-scattering sky (reference source/main.cpp:242-312, 647-757)."""
import ctypes as C

import numpy as np

F = np.float32


def _ray_sphere(o, d, radius):
    A = (d * d).sum(-1)
    B = 2 * (d * o).sum(-1)
    Cc = (o * o).sum(-1) - radius * radius
    disc = B * B - 4 * A * Cc
    ok = disc >= 0
    sq = np.sqrt(np.maximum(disc, 0))
    q = np.where(B < 0, -0.5 * (B - sq), -0.5 * (B + sq))
    with np.errstate(all="ignore"):
        x1 = q / A
        x2 = Cc / q
    t0 = np.minimum(x1, x2)
    t1 = np.maximum(x1, x2)
    return ok, t0, t1


def _host_sky(dirs, az_deg, el_deg, intensity):
    """float64 restatement of main.cpp:242-312 for an array of directions."""
    az = np.deg2rad(np.clip(az_deg, 0, 360)); el = np.deg2rad(90 - np.clip(el_deg, 0, 90))
    sun = np.array([np.sin(el) * np.cos(az), np.cos(el), np.sin(el) * np.sin(az)])
    sun /= np.linalg.norm(sun)
    Re, Ra, Hr, Hm = 6360e3, 6420e3, 7994.0, 1200.0
    bR = np.array([3.8e-6, 13.5e-6, 33.1e-6]); bM = np.array([21e-6] * 3)
    n = dirs.shape[0]
    pos = np.tile(np.array([0.0, 1000 + 6360e3, 0.0]), (n, 1))
    tmax = np.full(n, np.finfo(np.float32).max, np.float64)
    ok, t0, t1 = _ray_sphere(pos, dirs, Re)
    hit = ok & (t1 > 0)
    tmax = np.where(hit, np.maximum(0, t0), tmax)
    ok, t0, t1 = _ray_sphere(pos, dirs, Ra)
    tmin = np.where((t0 > 0), t0, 0.0)
    tmax = np.minimum(tmax, t1)
    seg = (tmax - tmin) / 16
    mu = dirs @ sun
    phaseR = 3 / (16 * np.pi) * (1 + mu * mu)
    g = 0.76
    phaseM = 3 / (8 * np.pi) * ((1 - g * g) * (1 + mu * mu)) / ((2 + g * g) * (1 + g * g - 2 * g * mu) ** 1.5)
    sumR = np.zeros((n, 3)); sumM = np.zeros((n, 3))
    odR = np.zeros(n); odM = np.zeros(n)
    tc = tmin.copy()
    for i in range(16):
        sp = pos + (tc + seg * 0.5)[:, None] * dirs
        h = np.linalg.norm(sp, axis=1) - Re
        hr = np.exp(-h / Hr) * seg; hm = np.exp(-h / Hm) * seg
        odR += hr; odM += hm
        _, _, t1l = _ray_sphere(sp, np.tile(sun, (n, 1)), Ra)
        segl = t1l / 8
        olR = np.zeros(n); olM = np.zeros(n); alive = np.ones(n, bool); tcl = np.zeros(n)
        for j in range(8):
            spl = sp + (tcl + segl * 0.5)[:, None] * sun
            hl = np.linalg.norm(spl, axis=1) - Re
            alive &= ~(hl < 0)
            olR += np.where(alive, np.exp(-hl / Hr) * segl, 0); olM += np.where(alive, np.exp(-hl / Hm) * segl, 0)
            tcl += segl
        tau = bR * (odR + olR)[:, None] + bM * 1.1 * (odM + olM)[:, None]
        att = np.exp(-tau) * alive[:, None]
        sumR += att * hr[:, None]; sumM += att * hm[:, None]
        tc += seg
    return (sumR * bR * phaseR[:, None] + sumM * bM * phaseM[:, None]) * intensity


def test_env_cdf_tables_match_numpy_restatement(pkg):
    lib = pkg.load_library()
    kp = pkg.abi.KernelParams()
    lib.vpt_kernel_params_default(C.byref(kp))
    kp.azimuth, kp.elevation = 75.0, 20.0
    kp.sky_color = pkg.abi.Float3(1.0, 0.9, 0.8)
    res = 48
    T = pkg.host.env_cdf_build(kp, res=res)
    el = np.arange(res) / (res - 1) * np.pi
    az = np.arange(res) / (res - 1) * np.pi * 2
    E, A = np.meshgrid(el, az, indexing="ij")
    dirs = np.stack([np.sin(E) * np.cos(A), np.cos(E), np.sin(E) * np.sin(A)], -1).reshape(-1, 3)
    val = _host_sky(dirs, 75.0, 20.0, np.array([1.0, 0.9, 0.8])).reshape(res, res, 3)
    np.testing.assert_allclose(T["val"][..., :3], val, rtol=3e-3, atol=1e-7)
    assert (T["val"][..., 3] == 1).all()
    func = np.linalg.norm(val, axis=-1)
    np.testing.assert_allclose(T["func"], func, rtol=3e-3, atol=1e-7)
    # cdf: running sum of the PREVIOUS texel's func / res, restarted per row, then normalised by the
    # row total and forced to 1 in the last column (main.cpp:688-731)
    f = T["func"].astype(np.float64).reshape(-1)
    prev = np.concatenate([[0.0], f[:-1]]).reshape(res, res)
    raw = np.cumsum(prev / res, axis=1)
    mfunc = raw[:, -1]
    np.testing.assert_allclose(T["marginal_func"], mfunc, rtol=1e-4)
    cdf = raw / mfunc[:, None]
    cdf[:, -1] = 1.0
    np.testing.assert_allclose(T["cdf"], cdf, rtol=2e-4, atol=1e-6)
    mc = np.cumsum(mfunc / res)
    assert T["marginal_int"] == np.float32(T["marginal_int"]) and abs(T["marginal_int"] - mc[-1]) <= 1e-4 * mc[-1]
    np.testing.assert_allclose(T["marginal_cdf"], mc / mc[-1], rtol=1e-4)
    assert T["marginal_cdf"][-1] == 1.0 and (np.diff(T["marginal_cdf"]) >= 0).all() and (np.diff(T["cdf"], axis=1) >= -1e-7).all()


def test_env_cdf_black_sky_falls_back_to_product_cdf(pkg):
    lib = pkg.load_library()
    kp = pkg.abi.KernelParams()
    lib.vpt_kernel_params_default(C.byref(kp))
    kp.sky_color = pkg.abi.Float3(0, 0, 0)
    T = pkg.host.env_cdf_build(kp, res=16)
    x = np.arange(16, dtype=np.float32) / 16
    np.testing.assert_array_equal(T["cdf"], x[None, :] * x[:, None])       # main.cpp:713-719
    assert T["marginal_int"] == 0.0 and T["marginal_cdf"][0] == 1.0


# ---- the pin on reference code: main.cpp's own lines compiled for the host (oracle/_ref/ref_env_cdf) -------------------
def _product_tables(pkg, az, el, sky):
    lib = pkg.load_library()
    kp = pkg.abi.KernelParams()
    lib.vpt_kernel_params_default(C.byref(kp))
    kp.azimuth, kp.elevation = float(az), float(el)
    kp.sky_color = pkg.abi.Float3(*[float(c) for c in sky])
    return pkg.host.env_cdf_build(kp, res=180)


def _assert_tables_identical(T, R):
    assert T["res"] == 180
    np.testing.assert_array_equal(T["val"][..., :3], R["val"])
    for k in ("func", "cdf", "marginal_func", "marginal_cdf"):
        np.testing.assert_array_equal(T[k], R[k], err_msg=k)
    assert np.float32(T["marginal_int"]) == np.float32(R["marginal_int"])


def test_env_cdf_bit_identical_to_reference_golden(pkg):
    """vpt_env_cdf_build == the tables written by the reference's own code (fixture from tests/golden/make_ref_env_cdf_golden.py)"""
    g = pkg.scene.load_golden("ref_env_cdf.npz")
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) == 3
    for name in names:
        az, el, r, gc, b = [float(x) for x in g[name + "/params"]]
        T = _product_tables(pkg, az, el, (r, gc, b))
        np.testing.assert_array_equal(T["val"][::6, :, :3], g[name + "/val_rows_every_6"])
        for k in ("func", "cdf", "marginal_func", "marginal_cdf"):
            np.testing.assert_array_equal(T[k], g[name + "/" + k], err_msg=k)
        assert np.float32(T["marginal_int"]) == np.float32(g[name + "/marginal_int"])


def test_env_cdf_bit_identical_to_reference_live(pkg):
    """the same against the program itself, on further suns (built where /root/reference exists; the GPU box gets it prebuilt)"""
    import pytest
    import ref_binding
    if not ref_binding.have_ref_env_cdf():
        pytest.skip("oracle/_ref/ref_env_cdf not available (no reference tree here)")
    for az, el, sky in ((120.0, 30.0, (1, 1, 1)), (0.0, 0.0, (1, 1, 1)), (359.0, 89.0, (2.0, 0.5, 0.25)), (200.0, -10.0, (1, 1, 1)), (33.0, 61.0, (0, 0, 0))):
        _assert_tables_identical(_product_tables(pkg, az, el, sky), ref_binding.ref_env_cdf(az, el, sky))
