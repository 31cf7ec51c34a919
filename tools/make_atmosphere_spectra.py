#!/usr/bin/env python3
"""Writes volumetric-path-tracer_amd/data/atmosphere_spectra.bin: the published data tables behind the reference's sky model,
read from the reference tree (runs only where /root/reference exists; the binary is committed):
   * solar irradiance 360..830 nm in 10 nm steps (ASTM G-173 based, via Bruneton's 2017 demo) -- atmosphere.h:66
   * ozone absorption cross-section, same sampling                                             -- atmosphere.h:75
   * CIE 1931 2-degree colour matching functions, 360..830 nm in 5 nm steps (95 rows x 4)     -- constants.h:71
   * the XYZ -> linear sRGB matrix                                                             -- constants.h:172
Data, not code: vpt_atmosphere_model (csrc/vpt_atmosphere.hip) loads it for every model other than the built-in default.
Layout (little endian): "VPTSPEC1" | i32 n=48, lambda_min=360, step=10 | f64 solar[n] | f64 ozone[n] | i32 rows=95 |
f64 cie[rows*4] | f64 xyz2srgb[9]."""
import os
import re
import struct

import numpy as np

SRC = "/root/reference/source/atmosphere/"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "volumetric-path-tracer_amd", "data", "atmosphere_spectra.bin")


def table(text, name):
    body = re.search(name + r"\[\d+\]\s*=\s*\{(.*?)\};", text, re.S).group(1)
    return np.array([float(x) for x in body.replace("\n", " ").split(",") if x.strip()], np.float64)


if __name__ == "__main__":
    h = open(SRC + "atmosphere.h").read()
    c = open(SRC + "constants.h").read()
    solar, ozone = table(h, "kSolarIrradiance"), table(h, "kOzoneCrossSection")
    cie, m = table(c, "CIE_2_DEG_COLOR_MATCHING_FUNCTIONS"), table(c, "XYZ_TO_SRGB")
    assert solar.size == 48 and ozone.size == 48 and cie.size == 380 and m.size == 9
    with open(OUT, "wb") as f:
        f.write(b"VPTSPEC1" + struct.pack("<iii", 48, 360, 10) + solar.astype("<f8").tobytes() + ozone.astype("<f8").tobytes() +
                struct.pack("<i", 95) + cie.astype("<f8").tobytes() + m.astype("<f8").tobytes())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
