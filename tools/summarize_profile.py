#!/usr/bin/env python3
"""Condenses what tools/profile_bench.sh left under gpurun_out/prof_<tag>/ into profiles/<tag>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats) and pmc_summary.json (per-launch means of the PMC
passes for the trace kernel plus derived figures).  Usage: python tools/summarize_profile.py <tag> [samples_per_launch]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    samples = float(sys.argv[2]) if len(sys.argv) > 2 else 1920 * 1080 * 16 * 64
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
    out = {}
    meta = None
    for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
        for f in glob.glob(os.path.join(src, d, "**", "*_counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "rt_trace" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
            for k, v in agg.items():
                out[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    if meta:
        out["_dispatch"] = {"vgpr": meta[0], "sgpr": meta[1], "lds_block": meta[2], "grid": meta[3], "workgroup": meta[4]}
    v = {k: x["mean_per_launch"] for k, x in out.items() if not k.startswith("_")}
    derived = {}
    if "SQ_INSTS_VALU" in v:
        derived["valu_insts_per_wave_sample"] = v["SQ_INSTS_VALU"] / (samples / 64.0)
        if "SQ_THREAD_CYCLES_VALU" in v:
            derived["mean_active_lanes_per_valu_inst"] = v["SQ_THREAD_CYCLES_VALU"] / v["SQ_INSTS_VALU"]
    if "GRBM_GUI_ACTIVE" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
        derived["gpu_cycles_per_launch"] = cyc
        if "SQ_INSTS_VALU" in v:
            derived["simd_cycles_per_valu_inst"] = 1024.0 * cyc / v["SQ_INSTS_VALU"]
        if "SQ_WAVE_CYCLES" in v:
            derived["mean_waves_per_simd"] = v["SQ_WAVE_CYCLES"] * 4.0 / cyc / 1024.0   # counter ticks every 4 cycles
    if "SQ_WAVE_CYCLES" in v:
        for name, key in (("wave_time_active", "SQ_ACTIVE_INST_ANY"), ("wave_time_wait_issue", "SQ_WAIT_INST_ANY"),
                          ("wave_time_wait_cnt", "SQ_WAIT_ANY")):
            if key in v:
                derived[name] = v[key] / v["SQ_WAVE_CYCLES"]
    # TCC counters: FETCH_SIZE counts 32-byte... the guide's gfx950 correction: per-XCD sums in kilobytes, reads x2
    if "FETCH_SIZE" in v:
        derived["hbm_read_bytes_per_launch"] = v["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in v:
        derived["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024.0
    out["_derived"] = derived
    json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
