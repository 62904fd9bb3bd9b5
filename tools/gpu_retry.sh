#!/bin/bash
# gpurun with retries while no GPU slot is free (exit code 3: nothing charged).  usage: tools/gpu_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for attempt in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 45
done
exit 3
