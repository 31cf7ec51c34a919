"""Build-time guard on the tracer kernels (no GPU needed: the code objects inside libvpt_hip.so are disassembled).

The timed instantiations of trace_kernel / trace_vol_kernel run at four waves per SIMD (128 VGPRs) WITHOUT scratch, and their launch constants
either sit in scalar registers or are re-read by scalar loads -- not parked in VGPR lanes and fetched back with v_readlane in the walk step
(tracer kernels; profiles/r04_four_waves.txt (j), (n): one spilled dword in the hot step costs 1-3 % of tracer time, the lane-parked
look-up matrix cost 4-8 %).  Both properties are the register allocator's decisions and a harmless-looking edit can flip them, so they are pinned."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "volumetric-path-tracer_amd", "libvpt_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def _code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        i = data.find(b"\x7fELF", pos)
        if i < 0:
            return
        pos = i + 4
        if data[i + 18:i + 20] != b"\xe0\x00":           # e_machine == EM_AMDGPU
            continue
        shoff, = struct.unpack_from("<Q", data, i + 0x28)
        shentsize, shnum = struct.unpack_from("<HH", data, i + 0x3A)
        end = i + shoff + shentsize * shnum
        yield data[i:end]
        pos = end


@pytest.fixture(scope="module")
def tracer_kernels(pkg):
    """{mangled kernel name: (metadata dict, instruction mnemonics)} of every trace_kernel / trace_vol_kernel instantiation"""
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    out = {}
    for blob in _code_objects(LIB):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(blob)
        try:
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
            if "trace_kernel" not in notes and "trace_vol_kernel" not in notes:
                continue
            meta = {}
            for blk in notes.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, None])[1]
                meta[g("name")] = {"vgpr": int(g("vgpr_count")), "scratch": int(g("private_segment_fixed_size")), "spill": int(g("vgpr_spill_count"))}
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        finally:
            os.unlink(f.name)
        cur = None
        for line in dis.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
            if m:
                cur = m.group(1) if ("trace_kernel" in m.group(1) or "trace_vol_kernel" in m.group(1)) else None
                if cur:
                    out[cur] = (meta.get(cur), [])
                continue
            if cur:
                m = re.match(r"^\s+([a-z][a-z0-9_]+)", line)
                if m:
                    out[cur][1].append(m.group(1))
    assert out, "no tracer kernels found in " + LIB
    return out


def _timed(name):
    """the non-counting instantiations that run at four waves: template argument COUNT is the 4th of trace_kernel<MULTI, COLOR, EMIT, COUNT, A24> and of
    trace_vol_kernel<MULTI, COLOR, EMIT, COUNT, SKYLUT, A24>; the SKYLUT ones (the Bruneton sky evaluated inside the tracer) run at two waves
    with 256 registers (vpt_trace_vol.hip) and are not held to this"""
    m = re.search(r"trace_(vol_)?kernelI((?:Lb[01]E)+)", name)
    if m is None:
        return False
    args = re.findall(r"Lb([01])E", m.group(2))
    return args[3] == "0" and not (m.group(1) and args[4] == "1")


def test_timed_tracer_kernels_fit_four_waves_without_scratch(tracer_kernels):
    timed = {k: v for k, v in tracer_kernels.items() if _timed(k)}
    assert len(timed) >= 20
    for name, (meta, ops) in timed.items():
        assert meta is not None, name
        assert meta["vgpr"] <= 128, (name, meta)
        assert meta["spill"] == 0, (name, meta)
        # (some instantiations reserve a few dozen bytes of private segment that no instruction touches -- a dead frame object of the backend;
        # what is pinned is that nothing is stored to or loaded from scratch)
        assert not [o for o in ops if o.startswith("scratch_") or o.startswith("buffer_")], name


def test_timed_tracer_kernels_do_not_park_launch_constants_in_vgpr_lanes(tracer_kernels):
    seen = 0
    for name, (_, ops) in tracer_kernels.items():
        if not _timed(name):
            continue
        seen += 1
        lanes = sum(1 for o in ops if o in ("v_readlane_b32", "v_writelane_b32"))
        args = re.findall(r"Lb([01])E", re.search(r"kernelI((?:Lb[01]E)+)", name).group(1))
        multi, emit = args[0] == "1", args[2] == "1"
        # measured: 1-15 of ~4700-5800 instructions (every single-volume instantiation, the instanced ones without an emission grid: the four
        # BASELINE configs' kernels among them); the instanced + emission instantiations still park ~170-290 (330-470 before the descriptors
        # were read per look-up)
        assert lanes <= (320 if (multi and emit) else 40), (name, lanes, len(ops))
    assert seen >= 20


def test_timed_raygen_kernels_do_not_spill(pkg):
    """raygen's timed instantiations (template argument COUNT = false) run without scratch: the closed-lens one at seven waves per SIMD (<= 72 registers), the
    open-lens one that resolves untraced samples from the lens domes at six (<= 80): at seven it kept 72 registers and spilled 7 dwords once its footprint became a
    square (round 5; config 5's raygen +5 %, repaired in round 6 -- profiles/r06_raygen.txt)"""
    if not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("no llvm-readelf in this image")
    seen = {}
    for blob in _code_objects(LIB):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(blob)
        try:
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        finally:
            os.unlink(f.name)
        if "raygen_kernel" not in notes:
            continue
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, None])[1]
            name = g("name")
            m = re.search(r"raygen_kernelILb([01])ELi(\d+)ELb([01])E", name or "")
            if m and m.group(1) == "0":
                seen[name] = (m.group(3) == "1", int(g("vgpr_count")), int(g("private_segment_fixed_size")), int(g("vgpr_spill_count")))
    assert len(seen) == 4, sorted(seen)                 # rows 16 / 64 x closed / open lens
    for name, (lens, vgpr, scratch, spill) in seen.items():
        assert scratch == 0 and spill == 0, (name, vgpr, scratch, spill)
        assert vgpr <= (80 if lens else 72), (name, vgpr)
