#!/bin/bash
# Runs on the GPU box (via gpurun): one rocprofv3 --pmc pass over a short bench.py run, then the per-kernel
# average of every counter.  usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]   (one pass = one run)
set -e
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-same-buffer > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
