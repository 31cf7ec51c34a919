#!/usr/bin/env python3
"""Print DESIGN 4's results table rows from a bench detail record (gpurun_out/bench_detail.json or profiles/r06_bench_detail.json):
    python tools/design_table.py profiles/r06_bench_detail.json"""
import json
import sys

d = json.load(open(sys.argv[1]))


def row(name, o, parity):
    r = o["roofline"]
    p = r["per_sample"]
    b = r["bytes_per_sample"]
    v = (r.get("valu") or {})
    k = (v.get("kernels") or {}).get(r["kernel"], {})
    frac = "void (`B` = %.0f: %.1fx the peak)" % (b["survey_8d_reference_counts"], r.get("reference_count_bytes_over_peak", 0)) if r["frac_void"] else "**%.3f** (%.1f)" % (r["frac"], b["survey_8d_reference_counts"])
    return "| %s | **%.2f** | %.2f (%.2f + %.2f + %.2f) | %.2f/%.2f/%.2f (%.2f/%.2f/%.2f) | %s | %.3f (%.1f + %.1f) | %s, %s | %.1f | %s | %s |" % (
        name, o["value"] / 1e3, o["ms_per_step"], r["raygen_ms_per_step"], r["trace_ms_per_step"], r["tail_resolve_ms_per_step"],
        p["density_fetches"], p["color_fetches"], p["emission_fetches"], p["density_lookups_reference"], p["color_lookups_reference"], p["emission_lookups_reference"],
        frac, r["frac_kernel_issued_fetches"], b["lookup_bytes"], b["record_stream_bytes"],
        ("%.1f" % r["traffic"]) if r.get("traffic") else "-", ("%.3f" % r["hbm_measured_frac"]) if r.get("hbm_measured_frac") else "-",
        r["tracer_grays_per_s"],
        ("%.3f (%.3f x %.1f)" % (v.get("useful_lane_issue", 0), k.get("valu_issue_busy_static_mix", 0), k.get("active_lanes_per_valu_instruction", 0))) if v else "-",
        parity)


cb = d.get("cpu_baseline") or {}
print(row("c2 (headline)", d, "vs the reference's kernel on %s host cores (%.1f Msamples/s), %s: %.1e, depth pixels differing %s" % (
    cb.get("cores"), cb.get("value", 0), (cb.get("sample") or "")[:30], cb.get("parity_rel_l2", 0), cb.get("parity_depth_pixels_differing"))))
for o in d.get("other_configs", []):
    par = o.get("parity") or {}
    print(row(o.get("name", "?"), o, "oracle, every %sth pixel, %s iterations: %.1e, depth differing %s" % (par.get("pixel_step"), par.get("iterations"), par.get("rel_l2", 0), par.get("depth_pixels_differing"))))
r = d["roofline"]
print("cold view: %.2f ms = %.1f Gsamples/s; cache build %.2f ms" % (r["cold_view_ms_per_step"], r["cold_view_msamples_per_s"] / 1e3, r["cache_build_ms_per_view"]))
pf = d.get("per_frame") or {}
print("per frame: %.4f ms = %.1f Gsamples/s; frame by frame %.4f ms = %.1f" % (pf.get("ms_per_frame", 0), pf.get("value", 0) / 1e3, (pf.get("frame_by_frame") or {}).get("ms_per_frame", 0), (pf.get("frame_by_frame") or {}).get("value", 0) / 1e3))
c1 = d.get("c1_cpu_single_thread") or {}
if c1:
    print("c1: cpu %.3f Msamples/s (%.1f s), hip %.1f Msamples/s (%.3f ms), parity %.1e" % (c1["cpu"]["value"], c1["cpu"]["seconds"], c1["hip"]["value"], c1["hip"]["ms_per_step"], c1["parity_rel_l2"]))
