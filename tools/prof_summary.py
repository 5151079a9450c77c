#!/usr/bin/env python3
"""Digest rocprofv3 output directories into one small JSON (+ markdown table) for profiles/.

  python tools/prof_summary.py --out profiles/r01_1080p_to_4k --stats DIR [--pmc DIR ...] [--note TEXT]

  --stats : a `rocprofv3 --kernel-trace --stats` output dir (…_kernel_stats.csv, …_kernel_trace.csv)
  --pmc   : any number of `rocprofv3 --pmc …` output dirs (…_counter_collection.csv), one per counter pass

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are
reported in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B, so it
is doubled (`fetch_bytes_corrected`); WRITE_SIZE matched the known byte count of our stores exactly
(64800 KiB for one 3840x2160 RGBA16F image) and is used as is.  Only kernels of this repository
(namespace fsr1) are kept.
"""
import argparse
import collections
import csv
import glob
import json
import os


def kernel_key(name):
    return name.split("(")[0].replace("void ", "").strip()


def _dbs(d):
    return glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)


def read_stats_db(d):
    """rocpd (sqlite) output of rocprofv3: per-kernel duration statistics from the `kernels` view."""
    import sqlite3
    import statistics
    out = {}
    for f in _dbs(d):
        c = sqlite3.connect(f)
        rows = collections.defaultdict(list)
        meta = {}
        for name, dur, gx, wx, lds, scr, vg, ag, sg in c.execute(
                "select name, duration, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels"):
            if "fsr1::" not in name:
                continue
            k = kernel_key(name)
            rows[k].append(dur)
            meta.setdefault(k, {"vgpr": vg, "agpr": ag, "sgpr": sg, "lds_bytes": lds, "scratch": scr, "workgroup": wx, "grid": gx})
        total = sum(sum(v) for v in rows.values()) or 1
        for k, v in rows.items():
            out[k] = {"calls": len(v), "avg_us": round(sum(v) / len(v) / 1e3, 3), "min_us": round(min(v) / 1e3, 3),
                      "max_us": round(max(v) / 1e3, 3), "stddev_us": round(statistics.pstdev(v) / 1e3, 3),
                      "percent_of_fsr1_time": round(100.0 * sum(v) / total, 2)}
            out[k].update(meta[k])
    return out


def read_pmc_db(d):
    import sqlite3
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in _dbs(d):
        c = sqlite3.connect(f)
        for name, ctr, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
            if "fsr1::" in name:
                agg[kernel_key(name)][ctr].append(float(val))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def read_stats(d):
    if _dbs(d):
        return read_stats_db(d)
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "fsr1::" not in r["Name"]:
                continue
            out[kernel_key(r["Name"])] = {
                "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3),
                "min_us": round(float(r["MinNs"]) / 1e3, 3), "max_us": round(float(r["MaxNs"]) / 1e3, 3),
                "stddev_us": round(float(r["StdDev"]) / 1e3, 3), "percent_of_gpu_time": float(r["Percentage"]),
            }
    res = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "fsr1::" not in r["Kernel_Name"]:
                continue
            k = kernel_key(r["Kernel_Name"])
            if k not in res:
                res[k] = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                          "lds_bytes": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"]),
                          "workgroup": int(r["Workgroup_Size_X"]), "grid": int(r["Grid_Size_X"])}
    for k, v in res.items():
        out.setdefault(k, {}).update(v)
    return out


def read_pmc(d):
    if _dbs(d):
        return read_pmc_db(d)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "fsr1::" not in r["Kernel_Name"]:
                continue
            agg[kernel_key(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True, help="output path prefix (writes PREFIX.json and PREFIX.md)")
    ap.add_argument("--stats", required=True)
    ap.add_argument("--pmc", action="append", default=[])
    ap.add_argument("--note", default="")
    ap.add_argument("--command", default="")
    ap.add_argument("--workload", default="", help="bench.py --workload this profile belongs to (lets bench.py find its PMC traffic)")
    ap.add_argument("--pipeline", default="", help="bench.py --pipeline this profile belongs to")
    ap.add_argument("--math", default="f")
    ap.add_argument("--storage", default="rgba16f")
    ap.add_argument("--lib", default="", help="the shared library profiled: per-kernel VGPR / SGPR / static LDS / scratch are read from its code objects")
    ap.add_argument("--bench-line", default="", help="file holding the JSON line bench.py printed during the kernel-trace pass")
    a = ap.parse_args()

    kernels = read_stats(a.stats)
    for d in a.pmc:
        for k, cs in read_pmc(d).items():
            kernels.setdefault(k, {}).setdefault("pmc_avg_per_launch", {}).update({c: round(v, 3) for c, v in cs.items()})
    for k, v in kernels.items():
        p = v.get("pmc_avg_per_launch", {})
        hbm = {}
        if "FETCH_SIZE" in p:
            hbm["fetch_bytes_raw"] = int(p["FETCH_SIZE"] * 1024)
            hbm["fetch_bytes_corrected"] = int(p["FETCH_SIZE"] * 1024 * 2)  # gfx950: 128-B requests tallied at 64 B
        if "WRITE_SIZE" in p:
            hbm["write_bytes"] = int(p["WRITE_SIZE"] * 1024)
        if "fetch_bytes_corrected" in hbm and "write_bytes" in hbm:
            hbm["traffic_bytes"] = hbm["fetch_bytes_corrected"] + hbm["write_bytes"]
        if hbm:
            v["hbm_per_launch"] = hbm
        if "TCC_HIT_sum" in p and "TCC_MISS_sum" in p:
            v["l2_hit_rate"] = round(p["TCC_HIT_sum"] / max(p["TCC_HIT_sum"] + p["TCC_MISS_sum"], 1.0), 4)
        if "SQ_INSTS_VALU" in p and "SQ_WAVES" in p:
            v["valu_insts_per_wave"] = round(p["SQ_INSTS_VALU"] / max(p["SQ_WAVES"], 1.0), 1)
        if "SQ_INSTS_VALU" in p and "avg_us" in v:
            # wave-instructions per SIMD per microsecond: 1024 SIMDs; a plain v_fma_f32 stream peaks at ~1000/us (2.4 cyc @2.4 GHz)
            v["valu_wave_insts_per_simd_per_us"] = round(p["SQ_INSTS_VALU"] / 1024.0 / v["avg_us"], 1)
    if a.lib:
        import importlib.util
        spec = importlib.util.spec_from_file_location("kernel_meta", os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_meta.py"))
        km = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(km)
        meta = km.kernel_meta(a.lib)
        for k, v in kernels.items():
            if k in meta:
                v["profiler_reported"] = {x: v.get(x) for x in ("vgpr", "sgpr", "lds_bytes", "scratch") if x in v}
                v.update({"vgpr": meta[k]["vgpr"], "sgpr": meta[k]["sgpr"], "lds_static_bytes": meta[k]["lds_static_bytes"],
                          "scratch": meta[k]["scratch_bytes"], "vgpr_spills": meta[k]["vgpr_spills"], "registers_from": "code object metadata"})
    doc = {"command": a.command, "note": a.note, "workload": a.workload, "pipeline": a.pipeline, "math": a.math, "storage": a.storage, "kernels": kernels}
    try:
        import importlib
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        doc["source_hash"] = importlib.import_module("fidelityfx-fsr_amd._lib").source_hash()
    except Exception as e:  # the summary is still useful without it; bench.py will then treat the profile as stale
        doc["source_hash"] = None
        doc["source_hash_error"] = str(e)
    if a.bench_line and os.path.exists(a.bench_line):
        for line in open(a.bench_line):
            if line.startswith("{"):
                try:
                    doc["bench_line"] = json.loads(line)
                except ValueError:
                    pass
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out + ".json", "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    with open(a.out + ".md", "w") as f:
        f.write("# rocprofv3 summary: %s\n\n" % os.path.basename(a.out))
        if a.command:
            f.write("Command: `%s`\n\n" % a.command)
        if a.note:
            f.write(a.note + "\n\n")
        if doc.get("source_hash"):
            f.write("Kernel sources (csrc/ + include/) sha256[:16] = `%s`; VGPR / SGPR / static LDS from the code-object metadata of the profiled library.\n\n" % doc["source_hash"])
        f.write("| kernel | calls | avg us | min us | max us | VGPR | SGPR | static LDS B | fetch MB (x2 corrected) | write MB | L2 hit | VALU insts/wave |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k in sorted(kernels):
            v = kernels[k]
            h = v.get("hbm_per_launch", {})
            f.write("| `%s` | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |\n" % (
                k, v.get("calls", ""), v.get("avg_us", ""), v.get("min_us", ""), v.get("max_us", ""), v.get("vgpr", ""),
                v.get("sgpr", ""), v.get("lds_static_bytes", v.get("lds_bytes", "")),
                round(h["fetch_bytes_corrected"] / 1e6, 2) if "fetch_bytes_corrected" in h else "",
                round(h["write_bytes"] / 1e6, 2) if "write_bytes" in h else "",
                v.get("l2_hit_rate", ""), v.get("valu_insts_per_wave", "")))
    print("wrote", a.out + ".json", a.out + ".md")


if __name__ == "__main__":
    main()
