#!/bin/bash
# Local helper (runs in the build container, not on the GPU box): submits one gpurun call and, only
# when the pool reports "no box free right now" (exit code 3, nothing ran, nothing charged), waits
# and submits the same call again.  Any other exit code ends it.
# usage: tools/gpurun_wait.sh <timeout-seconds> '<command>'
T=$1; shift
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 150
done
exit 3
