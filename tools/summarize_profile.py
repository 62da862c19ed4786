"""Condenses the rocprofv3 CSVs written by tools/gpu_profile.sh into one markdown summary."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]


def find(sub, pat):
    g = glob.glob(f"{out}/{sub}/*/*{pat}") + glob.glob(f"{out}/{sub}/*{pat}")
    return g[0] if g else None


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


print(f"# rocprofv3 summary ({out})\n")
bl = f"{out}/bench_line.json"
try:
    d = json.loads(open(bl).read())
    print("bench line of the traced run: value", d["value"], d["unit"], "| ms_per_step", d["ms_per_step"],
          "| kernels_ms (HIP events)", d["kernels_ms"], "\n")
except Exception:
    pass

st = find("trace", "kernel_stats.csv")
avg_ns = {}
if st:
    print("## --kernel-trace --stats\n")
    print("| kernel | calls | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(st)):
        n = short(r["Name"])
        avg_ns[n] = float(r["AverageNs"])
        print(f"| {n} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
              f"{float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.1f} |")
    print()


def pmc(sub):
    f = find(sub, "counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return agg
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


fetch, write, tcc, sq = pmc("fetch"), pmc("write"), pmc("tcc"), pmc("sq")
print("## HBM traffic per launch (separate --pmc passes)\n")
print("FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB. On gfx950 FETCH_SIZE counts 64 B per 128-B request of a")
print("wide coalesced read stream (MI355X_MICROARCH.md, HBM), so the read side is also shown doubled.\n")
print("| kernel | FETCH_SIZE MB | x2 (gfx950 correction) MB | WRITE_SIZE MB | TCC hit rate |")
print("|---|---|---|---|---|")
traffic = {}
for k in sorted(set(fetch) | set(write)):
    if "rocclr" in k:
        continue
    f = sum(fetch[k].get("FETCH_SIZE", [0])) / max(1, len(fetch[k].get("FETCH_SIZE", [0]))) * 1024 / 1e6
    w = sum(write[k].get("WRITE_SIZE", [0])) / max(1, len(write[k].get("WRITE_SIZE", [0]))) * 1024 / 1e6
    h = tcc[k].get("TCC_HIT_sum", [0]); m = tcc[k].get("TCC_MISS_sum", [0])
    hr = sum(h) / max(1.0, sum(h) + sum(m))
    traffic[k] = {"fetch_mb_raw": f, "fetch_mb_x2": 2 * f, "write_mb": w, "traffic_bytes": (2 * f + w) * 1e6}
    print(f"| {k} | {f:.1f} | {2*f:.1f} | {w:.1f} | {hr:.2f} |")
print()
print("## SQ counters per launch\n")
for k, v in sq.items():
    if "rocclr" in k:
        continue
    print(f"- {k}: " + ", ".join(f"{c}={sum(x)/len(x):.3g}" for c, x in sorted(v.items())))


def logical(name):
    for key, val in (("score_filter_kernel", "score_filter"), ("group_nms_kernel", "nms"), ("cc_nms_kernel", "nms"),
                     ("mask_assemble_kernel", "mask_assemble"), ("mask_rle_kernel", "mask_rle")):
        if key in name:
            return val
    return None


kernels_us = {logical(k): v / 1e3 for k, v in avg_ns.items() if logical(k)}
traffic_bytes = {logical(k): v["traffic_bytes"] for k, v in traffic.items() if logical(k)}
json.dump({"avg_ns": avg_ns, "kernels_us": kernels_us, "traffic": traffic, "traffic_bytes": traffic_bytes},
          open(f"{out}/summary.json", "w"), indent=1)
