#!/bin/bash
# gpurun with a wait for a free slot: retries ONLY when gpurun says that no box / slot was free (exit code 3: nothing ran,
# nothing was charged).  Usage: tools/gpu.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 30); do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 90
done
exit 3
