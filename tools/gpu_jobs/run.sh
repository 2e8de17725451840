#!/bin/bash
# tools/gpu_jobs/run.sh <timeout-seconds> <log> <command...>: gpurun with retries while no slot is free (exit code 3: nothing ran,
# nothing was charged); any other outcome is final
t=$1; log=$2; shift 2
for try in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
