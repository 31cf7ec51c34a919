#!/usr/bin/env python3
"""Writes tests/golden/ref_env_cdf.npz from oracle/_ref/ref_env_cdf -- the reference's own create_cdf table fill and host
single-scattering sky (source/main.cpp:181-312, 647-757) compiled where they lie (oracle/Makefile, target ref).  Only runs
where /root/reference exists; the fixture lets tests/test_env_cdf.py pin vpt_env_cdf_build elsewhere."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_binding  # noqa: E402

CASES = {"default_120_30": (120.0, 30.0, (1.0, 1.0, 1.0)), "low_sun_tinted": (75.0, 2.5, (1.0, 0.9, 0.8)), "zenith_clamped": (400.0, 95.0, (0.5, 0.7, 1.0))}

if __name__ == "__main__":
    out = {}
    for name, (az, el, sky) in CASES.items():
        t = ref_binding.ref_env_cdf(az, el, sky)
        out[name + "/params"] = np.array([az, el, *sky], np.float32)
        out[name + "/val_rows_every_6"] = t["val"][::6]                 # the colour table: every 6th row keeps the fixture small
        for k in ("func", "cdf", "marginal_func", "marginal_cdf"):
            out[name + "/" + k] = t[k]
        out[name + "/marginal_int"] = np.float32(t["marginal_int"])
    np.savez_compressed(os.path.join(HERE, "ref_env_cdf.npz"), **out)
    print("wrote ref_env_cdf.npz:", os.path.getsize(os.path.join(HERE, "ref_env_cdf.npz")), "bytes")
