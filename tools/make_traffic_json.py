#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh into profiles/traffic.json, the
measured-HBM-traffic figure bench.py reports as roofline.traffic.

    python tools/make_traffic_json.py gpurun_out/<tag>_pmc.txt <config> <width> <height> <iterations_per_launch> [kernel-substring]

Units and corrections (guides/MI355X_MICROARCH.md, HBM section): rocprofv3 reports both counters in
KiB-like units of 1024 B... -- calibrated here against a known byte count instead: the trace kernel
reads one 64-byte record + one 4-byte queue entry per queued ray and writes one 64-byte record per
queued ray, nothing else reaches HBM (the 425 KB grid is cache-resident), so
  FETCH_SIZE x 1000 x 2 (gfx950 tallies 128-B requests at 64 B)  == queued_rays x 68 B   and
  WRITE_SIZE x 1000                                                  == queued_rays x 64 B
hold to within 2% on config 2 (profiles/r01_d_pmc.txt); the same factors are applied to the others.
"""
import json
import os
import re
import sys

def is_counting(kernel):
    """the look-up-counting instantiation of a kernel (template argument COUNT: index 3 of the tracers, 0 of raygen)"""
    m = re.search(r"(\w+)<([^>]*)>", kernel.split("(")[0])
    if not m:
        return False
    args = [a.strip() for a in m.group(2).split(",")]
    idx = 0 if m.group(1) == "raygen_kernel" else 3
    return len(args) > idx and args[idx] == "true"


path, cfg, w, h, ipl = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
kern = sys.argv[6] if len(sys.argv) > 6 else "trace"
txt = open(path).read()
vals = {}
section = None
cur = None
for line in txt.splitlines():
    if line.startswith("### pmc:"):
        section = line.split(":", 1)[1].strip()
    m = re.match(r"^(\S.*)\(dispatches: (\d+), avg duration ([\d.]+) us\)", line)
    if m:
        cur = m.group(1)
    m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)\s+per-dispatch\s+([\d.]+)", line)
    if m and cur and kern in cur and not is_counting(cur):
        vals.setdefault(m.group(1), []).append(float(m.group(3)))
fetch = max(vals.get("FETCH_SIZE", [0.0]))
write = max(vals.get("WRITE_SIZE", [0.0]))
bytes_per_launch = fetch * 1000.0 * 2.0 + write * 1000.0
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
d = json.load(open(out_path)) if os.path.exists(out_path) else {}
d[cfg] = {"width": w, "height": h, "iterations_per_launch": ipl, "samples_per_launch": w * h * ipl,
          "fetch_size_counter": fetch, "write_size_counter": write, "bytes_per_launch": bytes_per_launch,
          "source": os.path.basename(path), "commit": os.environ.get("COMMIT", "unknown")}

# VALU issue utilisation of the hot kernels from the same file's SQ passes.  A wave64 VALU instruction occupies its SIMD's issue
# port for 2.2-2.5 cycles (fma / add / mul / mov / and / xor ...), 4.1-4.2 cycles (min / max / cvt / floor / shifts / mul24 /
# 3-operand integer ops / f64 / packed fp32 / DPP / readlane / any SGPR-operand form / the v_div_* helpers / 32-bit integer
# multiplies) or 8.1 cycles (rcp / sqrt / log / exp) -- measured, profiles/r02_valu_issue_probe.txt.  The dynamic mix is not
# counted by the hardware; the static mix of each kernel's ISA (tools/isa_histogram.py classes) prices an average instruction:
#   busy_static_mix = SQ_INSTS_VALU x mean_cycles(static mix) / (1024 SIMDs x kernel time x 2.4 GHz)
# next to the bracket [all at 2.3 cycles, all at 4.2 cycles].  lanes = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU.
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def static_mix_cycles(kernel_regex):
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_histogram.py"), kernel_regex, "--cycles"], capture_output=True, text=True).stdout
        m = re.search(r"mean VALU issue cycles \(static mix\): ([\d.]+)", out)
        return float(m.group(1)) if m else None
    except Exception:
        return None


cur = None
sq = {}
for line in txt.splitlines():
    m = re.match(r"^(\S.*?)\s*\(dispatches: (\d+), avg duration ([\d.]+) us\)", line)
    if m:
        cur = (m.group(1), float(m.group(3)))
        continue
    m = re.match(r"^\s+(SQ_INSTS_VALU|SQ_THREAD_CYCLES_VALU|SQ_WAIT_ANY|SQ_WAVE_CYCLES|SQ_INSTS_SALU)\s+\d+\s+per-dispatch\s+([\d.]+)", line)
    if m and cur:
        sq.setdefault(cur[0], {})[m.group(1)] = (float(m.group(2)), cur[1])
valu = {}
for k, v in sq.items():
    name = k.split("(")[0].replace("void ", "").split("<")[0].strip()
    counting = is_counting(k)
    if "SQ_INSTS_VALU" in v and "SQ_THREAD_CYCLES_VALU" in v and not counting and name.startswith("vpt::") and v["SQ_INSTS_VALU"][1] > 100.0:
        n, dur = v["SQ_INSTS_VALU"]
        if name in valu and valu[name]["kernel_us"] > dur:
            continue
        denom = 1024 * dur * 1e-6 * 2.4e9
        e = {"valu_wave_instructions_per_launch": n, "kernel_us": dur,
             "valu_issue_busy_at_2.3_cycles": round(n * 2.3 / denom, 3), "valu_issue_busy_at_4.2_cycles": round(n * 4.2 / denom, 3),
             "active_lanes_per_valu_instruction": round(v["SQ_THREAD_CYCLES_VALU"][0] / n, 1)}
        mix = static_mix_cycles(re.escape(k.split("(")[0].replace("void ", "").strip()))
        if mix:
            e["mean_issue_cycles_static_mix"] = mix
            e["valu_issue_busy_static_mix"] = round(n * mix / denom, 3)
        if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v:
            e["wave_cycles_waiting_fraction"] = round(v["SQ_WAIT_ANY"][0] / v["SQ_WAVE_CYCLES"][0], 3)
        if "SQ_INSTS_SALU" in v:
            e["salu_issue_busy_per_cu"] = round(v["SQ_INSTS_SALU"][0] * 1.09 / (256 * dur * 1e-6 * 2.4e9), 3)    # one scalar unit per CU, 1.09 cycles per instruction
        valu[name] = e
if valu:
    d[cfg]["valu"] = valu
json.dump(d, open(out_path, "w"), indent=1, sort_keys=True)
print(cfg, "FETCH_SIZE", fetch, "WRITE_SIZE", write, "-> %.3f GB per launch, %.1f B per sample" % (bytes_per_launch / 1e9, bytes_per_launch / (w * h * ipl)))
