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
    if m and cur and kern in cur and "true>" not in cur.split("(")[0][-8:]:
        vals.setdefault(m.group(1), []).append(float(m.group(3)))
fetch = max(vals.get("FETCH_SIZE", [0.0]))
write = max(vals.get("WRITE_SIZE", [0.0]))
bytes_per_launch = fetch * 1000.0 * 2.0 + write * 1000.0
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
d = json.load(open(out_path)) if os.path.exists(out_path) else {}
d[cfg] = {"width": w, "height": h, "iterations_per_launch": ipl, "samples_per_launch": w * h * ipl,
          "fetch_size_counter": fetch, "write_size_counter": write, "bytes_per_launch": bytes_per_launch,
          "source": os.path.basename(path)}

# VALU issue utilisation of the three hot kernels from the same file's SQ passes: a wave64 VALU instruction occupies a
# SIMD for 4 cycles, so busy = SQ_INSTS_VALU x 4 / (1024 SIMDs x kernel time x 2.4 GHz); lanes = SQ_THREAD_CYCLES_VALU /
# SQ_INSTS_VALU (bench.py reports the dominant kernel's pair next to the HBM roofline)
cur = None
sq = {}
for line in txt.splitlines():
    m = re.match(r"^(\S.*?)\s*\(dispatches: (\d+), avg duration ([\d.]+) us\)", line)
    if m:
        cur = (m.group(1), float(m.group(3)))
        continue
    m = re.match(r"^\s+(SQ_INSTS_VALU|SQ_THREAD_CYCLES_VALU)\s+\d+\s+per-dispatch\s+([\d.]+)", line)
    if m and cur:
        sq.setdefault(cur[0], {})[m.group(1)] = (float(m.group(2)), cur[1])
valu = {}
for k, v in sq.items():
    name = k.split("(")[0].replace("void ", "").split("<")[0].strip()
    counting = "true>" in k.split("(")[0][-8:]
    if len(v) == 2 and not counting and name.startswith("vpt::") and v["SQ_INSTS_VALU"][1] > 100.0:
        n, dur = v["SQ_INSTS_VALU"]
        valu[name] = {"valu_wave_instructions_per_launch": n, "kernel_us": dur,
                      "valu_issue_busy": round(n * 4 / (1024 * dur * 1e-6 * 2.4e9), 3),
                      "active_lanes_per_valu_instruction": round(v["SQ_THREAD_CYCLES_VALU"][0] / n, 1)}
if valu:
    d[cfg]["valu"] = valu
json.dump(d, open(out_path, "w"), indent=1, sort_keys=True)
print(cfg, "FETCH_SIZE", fetch, "WRITE_SIZE", write, "-> %.3f GB per launch, %.1f B per sample" % (bytes_per_launch / 1e9, bytes_per_launch / (w * h * ipl)))
