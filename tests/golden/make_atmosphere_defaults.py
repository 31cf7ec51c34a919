#!/usr/bin/env python3
"""Derive the scalar constants of the reference's DEFAULT atmosphere model that need its data
tables (CIE 1931 2-degree colour matching functions, ozone cross-sections): run in the build
container, prints the numbers hard-coded in csrc/vpt_atmosphere.hip::default_model().

Reference: atmosphere::init / update_model / convert_spectrum_to_linear_srgb /
compute_spectral_radiance_to_luminance_factors (source/atmosphere/atmosphere.cpp:123-238,
698-784, 1177-1230) with m_use_constant_solar_spectrum = m_use_ozone = m_do_white_balance = true.
"""
import re
import numpy as np

SRC = "/root/reference/source/atmosphere/"
txt = open(SRC + "constants.h").read()
cie = re.search(r"CIE_2_DEG_COLOR_MATCHING_FUNCTIONS\[380\]\s*=\s*\{(.*?)\};", txt, re.S).group(1)
cie = np.array([float(x) for x in cie.replace("\n", " ").split(",") if x.strip()]).reshape(95, 4)
xyz2srgb = np.array([3.2406, -1.5372, -0.4986, -0.9689, 1.8758, 0.0415, 0.0557, -0.2040, 1.0570]).reshape(3, 3)
h = open(SRC + "atmosphere.h").read()
ozone = re.search(r"kOzoneCrossSection\[48\]\s*=\s*\{(.*?)\};", h, re.S).group(1)
ozone = np.array([float(x) for x in ozone.replace("\n", " ").split(",") if x.strip()])
LMIN, LMAX = 360, 830


def cmf(wl, col):
    if wl <= LMIN or wl >= LMAX:
        return 0.0
    u = (wl - LMIN) / 5.0
    row = int(np.floor(u))
    u -= row
    return cie[row, col] * (1.0 - u) + cie[row + 1, col] * u


wls = np.arange(LMIN, LMAX + 1, 10, dtype=float)
solar = np.full_like(wls, 1.5)


def interp(fn, wl):
    if wl < wls[0]:
        return fn[0]
    for i in range(len(wls) - 1):
        if wl < wls[i + 1]:
            u = (wl - wls[i]) / (wls[i + 1] - wls[i])
            return fn[i] * (1 - u) + fn[i + 1] * u
    return fn[-1]


x = y = z = 0.0
for lam in range(LMIN, LMAX):
    v = interp(solar, lam)
    x += cmf(lam, 1) * v; y += cmf(lam, 2) * v; z += cmf(lam, 3) * v
rgb = 683.0 * (xyz2srgb @ np.array([x, y, z]))
wp = rgb / rgb.mean()
print("white_point", ["%.9g" % v for v in wp])


def k_factors(power):
    k = np.zeros(3)
    lam_rgb = (680.0, 550.0, 440.0)
    s = [interp(solar, l) for l in lam_rgb]
    for lam in range(LMIN, LMAX):
        bar = xyz2srgb @ np.array([cmf(lam, 1), cmf(lam, 2), cmf(lam, 3)])
        irr = interp(solar, lam)
        for c in range(3):
            k[c] += bar[c] * irr / s[c] * (lam / lam_rgb[c]) ** power
    return k * 683.0


print("sky_k", ["%.9g" % v for v in k_factors(-3)])
print("sun_k", ["%.9g" % v for v in k_factors(0)])
dob = 2.687e20
maxoz = 300.0 * dob / 15000.0
absorption = maxoz * ozone
print("absorption_extinction@680,550,440", ["%.9g" % interp(absorption, l) for l in (680.0, 550.0, 440.0)])
