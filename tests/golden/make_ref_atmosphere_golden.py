"""Writes tests/golden/ref_atmosphere_sub.npz: every 8th texel (4th along r) of the four tables the REFERENCE'S OWN
precompute kernels produce for its default sky (oracle/_ref/libvptref_atm.so, see oracle/ref_shim/).  Run where
/root/reference exists:   make -C oracle ref && python tests/golden/make_ref_atmosphere_golden.py   (~80 s on 8 cores)
`... make_ref_atmosphere_golden.py luminance` writes ref_atmosphere_luminance_sub.npz: the same for use_luminance = PRECOMPUTED.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding  # noqa: E402
import test_gpu_atmosphere_vs_ref as T  # noqa: E402

luminance = len(sys.argv) > 1 and sys.argv[1] == "luminance"      # the PRECOMPUTED luminance mode's five passes (~8 min on 8 cores)
if luminance:
    ref, defined = T.reference_luminance_tables(oracle_binding.pkg, 4, **T.LUM_OPTIONS)
else:
    ref, defined = T.reference_tables(oracle_binding.pkg)
out = {k: np.ascontiguousarray(T.subsample(k, ref[k])) for k in T.NAMES}
out.update({k + "/defined": np.ascontiguousarray(T.subsample(k, defined[k])) for k in T.NAMES})
for k in T.NAMES:
    print(k, "texels independent of out-of-bounds reads: %.1f %%" % (100.0 * defined[k].mean()))
path = T.LUM_GOLDEN if luminance else os.path.join(HERE, "ref_atmosphere_sub.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes", {k: v.shape for k, v in out.items()})
