#!/bin/bash
# gpurun with retries while no slot is free (exit code 3): usage: scripts/gpurun_retry.sh <log> <timeout> '<command>'
log=$1; to=$2; cmd=$3
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$cmd" > "$log" 2>&1
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
