#!/usr/bin/env python3
"""Static opcode histogram of one kernel of the built library (which instruction classes does the hot loop consist of?).
    python tools/isa_histogram.py <demangled-name-regex> [lib.so]"""
import collections, os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
pat = re.compile(args[0])
so = args[1] if len(args) > 1 else os.path.join(ROOT, "volumetric-path-tracer_amd", "libvpt_hip.so")
WANT_CYCLES = "--cycles" in sys.argv


def issue_cycles(op):
    """measured issue cycles per wave64 instruction at 8 waves per SIMD (profiles/r02_valu_issue_probe.txt); None: not a VALU instruction"""
    if not op.startswith("v_"):
        return None
    o = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if o.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_log", "v_exp", "v_sin", "v_cos")):
        return 8.1
    cheap = ("v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
             "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_fmamk_f32", "v_fmaak_f32", "v_mul_legacy_f32", "v_cndmask_b32", "v_cmp", "v_accvgpr", "v_nop")
    if o.startswith(cheap) and not op.endswith("_e64") and not op.endswith("_dpp") and not op.endswith("_sdwa"):
        return 2.3
    if o.startswith("v_fma_f32") or o.startswith("v_mov_b32"):
        return 2.3
    return 4.2
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
        if WANT_CYCLES:
            tot_c, tot_n = 0.0, 0
            for op, n in c.items():
                cyc = issue_cycles(op)
                if cyc is not None:
                    tot_c += cyc * n
                    tot_n += n
            print("   mean VALU issue cycles (static mix): %.3f over %d VALU instructions" % (tot_c / max(1, tot_n), tot_n))
