#!/usr/bin/env python3
"""Static opcode histogram of one kernel of the built library (which instruction classes does the hot loop consist of?).
    python tools/isa_histogram.py <demangled-name-regex> [lib.so]"""
import collections, os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = re.compile(sys.argv[1])
so = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "volumetric-path-tracer_amd", "libvpt_hip.so")
data = open(so, "rb").read()
pos = 0
while True:
    i = data.find(b"\x7fELF", pos)
    if i < 0:
        break
    pos = i + 4
    if data[i + 18:i + 20] != b"\xe0\x00":
        continue
    shoff, = struct.unpack_from("<Q", data, i + 0x28)
    shentsize, shnum = struct.unpack_from("<HH", data, i + 0x3A)
    end = i + shoff + shentsize * shnum
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[i:end])
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--demangle", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    pos = end
    cur = None
    hist = collections.defaultdict(collections.Counter)
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^\s+(\S+)", line)
        if cur and m and pat.search(cur) and not cur.startswith("."):
            hist[cur][m.group(1)] += 1
    for k, c in hist.items():
        tot = sum(c.values())
        if tot < 50:
            continue
        print("== %s: %d instructions" % (k, tot))
        cls = collections.Counter()
        for op, n in c.items():
            if op.startswith("v_cmp"): cls["v_cmp*"] += n
            elif op.startswith("v_cndmask"): cls["v_cndmask"] += n
            elif op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_log", "v_exp", "v_sin", "v_cos")): cls["transcendental"] += n
            elif op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_u32", "v_mul_u32")): cls["int mul"] += n
            elif op.startswith(("v_div_", )): cls["v_div_*"] += n
            elif op.startswith("v_"): cls["other VALU"] += n
            elif op.startswith("s_"): cls["SALU/branch/wait"] += n
            elif op.startswith("ds_"): cls["LDS"] += n
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cls["VMEM"] += n
            else: cls["other"] += n
        print("   " + ", ".join("%s %d (%.0f%%)" % (a, b, 100.0 * b / tot) for a, b in cls.most_common()))
        print("   top: " + ", ".join("%s %d" % kv for kv in c.most_common(40)))
