#!/bin/bash
# One rocprofv3 --pmc pass (its own run, no tracing options) over bench.py; prints per-kernel averages.
# usage: tools/pmc_pass.sh <tag> <nms_mode> <counter> [<counter> ...]
set -e
export TMPDIR=/tmp
TAG=$1; MODE=$2; shift 2
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --nms-mode $MODE > "$OUT/run.log" 2>&1
python3 tools/pmc_summary.py "$OUT"
