#!/bin/bash
# Diagnostic (stamps build): phase 2 replaced by synthetic instruction mixes.  usage: tools/sf_ablate.sh <mode>
m=${1:-cc}
for abl in 6 7 8 9; do
    echo "== abl=$abl"
    YP_SF_ABL=$abl timeout -k 10 100 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --nms-mode $m 2>&1 | grep -E -o '"score_filter": [0-9.]*|\[yp stamps\] tiles.*' || exit 1
done
