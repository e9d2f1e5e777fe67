#!/bin/bash
# tools/gpurun_retry.sh TIMEOUT 'command': gpurun, retried only while it reports "no slot free" (exit code 3, nothing charged)
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
