"""Build libvpt_hip.so (the C-ABI library of include/vpt_abi.h) for gfx950 with hipcc.

    python volumetric-path-tracer_amd/build.py [--force] [--verbose] [--variant NAME [--with-pool] [--arith fast]] [-DNAME=VALUE ...]

Flags that matter for parity (Arithmetic): the tracer, the host code and the
resolve kernel are compiled STRICT: -ffp-contract=off (no FMA contraction on the decision
path), HIP's default correctly-rounded fp32 divide/sqrt, no fast-math.  Only vpt_tail.hip
(the value-only sky/environment tail) is compiled with approximate divide/sqrt and FMA
contraction.  The .so is built IN-TREE so it travels to the GPU box with the snapshot.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
OUT = os.path.join(HERE, "libvpt_hip.so")
# every header a source may include: all of csrc/*.h and include/*.h (a stale object after a header edit is how an old fast_div verdict
# or an old ABI struct would survive an incremental build)
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
          sorted(os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include")) if f.endswith(".h"))

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: the SLP vectoriser turns pairs of fp32 operations into packed v_pk_* instructions -- the same issue cost as the two
# scalar ones on gfx950 (profiles/r02_valu_issue_probe.txt) but operands in aligned VGPR pairs, constants included: the direct tracer needs
# 133 registers without them and 158 with (profiles/r04_four_waves.txt); every kernel of the path is at least as fast without.
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-fno-slp-vectorize"]
STRICT = ["-ffp-contract=off", "-fno-fast-math"]
VALUE_ONLY = ["-ffp-contract=off", "-fno-hip-fp32-correctly-rounded-divide-sqrt"]
SOURCES = {
    "vpt_host.hip": STRICT,
    "vpt_caches.hip": STRICT,
    "vpt_trace.hip": STRICT,
    "vpt_trace_vol.hip": STRICT,
    "vpt_resolve.hip": STRICT,
    "vpt_tail.hip": VALUE_ONLY,
    "vpt_atmosphere.hip": STRICT,
    "vpt_env.hip": STRICT,
    "vpt_io.hip": STRICT,
    "vpt_testhooks.hip": STRICT,
}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# study sources that are NOT part of the product library (--with-pool: the round-3 pool tracer, selected at run time by VPT_TRACER=pool)
POOL_SOURCE = os.path.join("variants", "vpt_trace_pool.hip")


# --arith fast (study builds only): the tracer compiled the way a renderer without a bit-parity contract would be -- FMA contraction, approximate
# divide / sqrt.  A DIAGNOSTIC of what the arithmetic contract costs; its images are close to, not equal to, the reference's.
FAST_ARITH = ["-ffp-contract=fast", "-fno-hip-fp32-correctly-rounded-divide-sqrt"]
FAST_ARITH_SOURCES = ("vpt_trace.hip", "vpt_trace_vol.hip")


def build(force=False, verbose=False, extra_flags=(), variant=None, with_pool=False, arith=None):
    """variant: build libvpt_hip_<variant>.so with `extra_flags` next to the default library (perf
    experiments: select it at run time with VPT_LIB_PATH)."""
    global OBJ, OUT
    sources = dict(SOURCES)
    if arith == "fast":
        if not variant:
            raise RuntimeError("--arith fast builds a study library: give it a --variant name")
        for src in FAST_ARITH_SOURCES:
            sources[src] = FAST_ARITH
    if with_pool:
        if not variant:
            raise RuntimeError("--with-pool builds a study library: give it a --variant name")
        sources[POOL_SOURCE] = STRICT
        extra_flags = list(extra_flags) + ["-DVPT_WITH_POOL"]
    if variant:
        # objects of a study build live outside the tree (nothing to clean up, nothing extra for gpurun to push)
        OBJ = os.path.join(os.environ.get("TMPDIR", "/tmp"), "vpt_obj_" + variant)
        OUT = os.path.join(HERE, "libvpt_hip_%s.so" % variant)
    os.makedirs(OBJ, exist_ok=True)
    common_deps = list(HEADERS) + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src, flags in sources.items():
        obj = os.path.join(OBJ, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        if force or (extra_flags and not variant) or _stale(obj, [os.path.join(CSRC, src)] + common_deps):
            jobs.append([HIPCC] + COMMON + flags + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            for cmd, r in ex.map(run, jobs):
                if r.returncode != 0:
                    sys.stderr.write(r.stdout + r.stderr)
                    raise RuntimeError("hipcc failed: " + " ".join(cmd))
                if verbose and r.stderr:
                    print(r.stderr)
    if jobs or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lz", "-o", OUT]
        cmd, r = run(cmd)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc link failed")
    if not variant:
        build_cli(verbose=verbose)
    return OUT


CLI_SRC = os.path.join(HERE, "..", "tools", "vpt_cli.cpp")
CLI_OUT = os.path.join(HERE, "vpt_cli")


def build_cli(verbose=False):
    """the headless main.cpp-equivalent (tools/vpt_cli.cpp), linked against libvpt_hip.so next to it"""
    inc = os.path.join(HERE, "..", "include")
    deps = [CLI_SRC, OUT, os.path.join(inc, "vpt_abi.h"), os.path.join(inc, "vpt_io.h")]
    if not _stale(CLI_OUT, deps):
        return CLI_OUT
    cmd = [HIPCC, "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-I", inc, CLI_SRC, "-o", CLI_OUT,
           "-L", HERE, "-lvpt_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building vpt_cli")
    return CLI_OUT


if __name__ == "__main__":
    variant = None
    argv = list(sys.argv[1:])
    if "--variant" in argv:
        i = argv.index("--variant")
        variant = argv[i + 1]
        del argv[i:i + 2]
    arith = None
    if "--arith" in argv:
        i = argv.index("--arith")
        arith = argv[i + 1]
        del argv[i:i + 2]
    build(force="--force" in argv, verbose="--verbose" in argv,
          extra_flags=[a for a in argv if a.startswith("-") and a not in ("--force", "--verbose", "--with-pool")], variant=variant,
          with_pool="--with-pool" in argv, arith=arith)
    print("built", OUT)
