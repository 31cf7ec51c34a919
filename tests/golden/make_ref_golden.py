"""Writes tests/golden/ref_golden.npz: the buffers the REFERENCE'S OWN KERNEL (oracle/_ref/libvptref.so =
/root/reference/source/render_kernel.cu compiled for the CPU, see oracle/ref_shim/) leaves on the scenes of
tests/ref_cases.py.  Run in a container that has /root/reference:

    make -C oracle ref && python tests/golden/make_ref_golden.py

The file lets tests/test_oracle_vs_ref.py pin the oracle where the reference does not exist.  Only accum, depth and
the alpha channel are stored (display/raw rgb are functions of accum, checked live).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_binding  # noqa: E402
import ref_cases  # noqa: E402


def main():
    out = {}
    for name in ref_cases.CASES:
        sd, iters = ref_cases.build(name)
        r = ref_binding.RefBinding(sd)
        r.render(iters)
        out[name + "/accum"] = r.accum.astype(np.float32).reshape(sd.height, sd.width, 3)
        out[name + "/depth"] = r.depth.astype(np.float32).reshape(sd.height, sd.width)
        out[name + "/alpha"] = r.raw[:, 3].astype(np.float32).reshape(sd.height, sd.width)
        out[name + "/display"] = r.display.reshape(sd.height, sd.width)
        print(name, "mean", float(r.accum.mean()), "hit pixels", int(np.count_nonzero(r.depth)))
    path = os.path.join(HERE, "ref_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
