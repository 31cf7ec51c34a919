"""Build libvpt_hip.so (the C-ABI library of include/vpt_abi.h) for gfx950 with hipcc.

    python volumetric-path-tracer_amd/build.py [--force] [--verbose]

Flags that matter for parity (Arithmetic): -ffp-contract=off (no FMA
contraction on the decision path), default correctly-rounded fp32 divide/sqrt, no
fast-math.  The .so is built IN-TREE so it travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvpt_hip.so")
SOURCES = ["vpt_host.hip", "vpt_trace.hip", "vpt_resolve.hip", "vpt_atmosphere.hip", "vpt_testhooks.hip"]
HEADERS = ["vpt_math.h", "vpt_device.h", "vpt_rng.h", os.path.join("..", "..", "include", "vpt_abi.h"),
           os.path.join("..", "..", "include", "vpt_testhooks.h")]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return OUT
    cmd = [HIPCC] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libvpt_hip.so")
    if verbose and (r.stdout or r.stderr):
        print(r.stdout + r.stderr)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra_flags=[a for a in sys.argv[1:] if a.startswith("-") and a not in ("--force", "--verbose")])
    print("built", OUT)
