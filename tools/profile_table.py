#!/usr/bin/env python3
"""Markdown table of the per-config profiles under profiles/ (the one in DESIGN.md section 5): python tools/profile_table.py r02"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
print("| profile (`profiles/%s_*`) | Mpix/s | µs per step | EASU µs (frac of 8 TB/s) | RCAS µs (frac) | fused µs (frac) |" % tag)
print("|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", tag + "_*.json"))):
    name = os.path.basename(f)[len(tag) + 1:-5]
    prof = json.load(open(f))
    line = prof.get("bench_line") if isinstance(prof, dict) else None  # (the bench line of the traced run travels inside the summary since round 6)
    if not line or "kernels" not in prof:
        continue  # (bench lines, parity reports: not profiles)
    h = line["config"].get("workload", "").find("math=h") >= 0

    def cell(kind, alg_key):
        want = {"easu": "easu_h_kernel" if h else "::easu_kernel<", "rcas": "rcas_h_kernel" if h else "::rcas_kernel<", "fused": "::fused_kernel<"}[kind]
        if kind not in line["kernels"]:
            return "—"
        # the pipeline's own variant of the kernel template is the one launched most (a trace may hold EXACT / generic ones too)
        hits = [v for k, v in prof["kernels"].items() if want in k or (kind == "fused" and "::fused_s2_kernel<" in k)]
        if not hits:
            return "—"
        v = max(hits, key=lambda x: x.get("calls", 0))
        alg = line["kernels"][kind]["algorithmic_bytes"]
        return "%.1f (%.2f)" % (v["avg_us"], alg / (v["avg_us"] * 1e-6) / 8e12)
    print("| `%s` | %s | %.1f | %s | %s | %s |" % (name, format(int(round(line["value"])), ","), line["ms_per_step"] * 1e3, cell("easu", 0), cell("rcas", 0), cell("fused", 0)))
