#!/bin/bash
# gpurun with a retry ONLY while no box / slot is free (exit code 3: nothing ran, nothing was charged)
# usage: gpurun_retry.sh <timeout> <command>
T=$1; shift
for k in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc = 3 ] || exit $rc
  sleep 75
done
exit 3
