#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + HBM traffic counters for bench.py.
# Each rocprofv3 pass is its own run (the PMC passes never share a run with tracing options).
set -e
export TMPDIR=/tmp
# usage: tools/gpu_profile.sh <tag> <per_class|cc> [extra bench.py arguments, e.g. --config 4]
OUT=gpurun_out/prof_${1:-r03}
MODE=${2:-per_class}
shift; shift || true
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --skip-same-buffer --nms-mode $MODE $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d "$OUT/tcc" -- $CMD > "$OUT/tcc.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/sq" -- $CMD > "$OUT/sq.log" 2>&1
grep -h '"metric"' "$OUT/trace.log" | tail -1 > "$OUT/bench_line.json" || true
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md"
cat "$OUT/summary.md"
