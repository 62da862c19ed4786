#!/usr/bin/env python3
"""Copies the judged files of tools/gpu_profile.sh runs from gpurun_out/ (scratch) into profiles/<round>/ and
refreshes profiles/hbm_traffic.json (the per-launch HBM bytes bench.py quotes as roofline.traffic).

    python tools/collect_profiles.py r03 gpurun_out/prof_r03_pc:per_class gpurun_out/prof_r03_cc:cc
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd, pairs = sys.argv[1], [a.split(":") for a in sys.argv[2:]]
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for src, mode in pairs:
        src = os.path.join(ROOT, src)
        shutil.copy(os.path.join(src, "summary.md"), os.path.join(dst, f"rocprof_summary_{mode}.md"))
        shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, f"rocprof_summary_{mode}.json"))
        shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, f"bench_line_traced_{mode}.json"))
        stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True))
        if stats:
            shutil.copy(stats[-1], os.path.join(dst, f"kernel_stats_{mode}.csv"))
        sj = json.load(open(os.path.join(src, "summary.json")))
        for k, v in sj.get("traffic_bytes", {}).items():
            traffic[f"{mode}:{k}"] = int(round(v))
    json.dump(traffic, open(tpath, "w"), indent=1)
    print("collected into", dst, "and", tpath)


if __name__ == "__main__":
    main()
