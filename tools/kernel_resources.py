#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every gfx950 kernel in a built library or object, and how many v_readlane / v_writelane / s_load instructions it
holds (launch constants parked in VGPR lanes: DESIGN 2, "scalar registers are a budget too"; pinned by tests/test_kernel_resources.py).
    python tools/kernel_resources.py [path/to/lib.so] [name-regex]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(ROOT, "volumetric-path-tracer_amd", "libvpt_hip.so")
filt = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
data = open(so, "rb").read()
rows = []
pos = 0
while True:
    i = data.find(b"\x7fELF", pos)
    if i < 0:
        break
    pos = i + 4
    if data[i + 18:i + 20] != b"\xe0\x00":           # e_machine == EM_AMDGPU (224)
        continue
    # section-header table end bounds the embedded ELF
    import struct
    shoff, = struct.unpack_from("<Q", data, i + 0x28)
    shentsize, shnum = struct.unpack_from("<HH", data, i + 0x3A)
    end = i + shoff + shentsize * shnum
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[i:end])
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    lanes = {}
    cur = None
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            cur = m.group(1)
            lanes[cur] = [0, 0, 0]
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", line) if cur else None
        if m:
            lanes[cur][0] += 1
            lanes[cur][1] += m.group(1) in ("v_readlane_b32", "v_writelane_b32")
            lanes[cur][2] += m.group(1).startswith("s_load")
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        rows.append((g("name"), g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count"),
                     lanes.get(g("name"), [0, 0, 0])))
    pos = end
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
    if filt.search(n):
        print("%-100s vgpr %4s agpr %3s sgpr %4s scratch %5s lds %6s spill %s  | instr %5d lane r/w %4d s_load %3d" % (n[:100], r[1], r[2], r[3], r[4], r[5], r[6], r[7][0], r[7][1], r[7][2]))
