#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (nothing is charged for those):  gpurun_retry.sh LOG TIMEOUT 'command'
LOG=$1; T=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  grep -q "status=transient" $LOG || exit 0
  sleep 45
done
