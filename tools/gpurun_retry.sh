#!/bin/bash
# tools/gpurun_retry.sh TIMEOUT 'command': gpurun, retried only while it reports "no slot free" (exit code 3, nothing charged)
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q '"status": "transient"' /root/repo/gpurun_out/.last_call.json 2>/dev/null; then exit $rc; fi
  [ $rc -ne 3 ] && [ $rc -ne 0 ] && exit $rc
  grep -q '"status": "transient"' /root/repo/gpurun_out/.last_call.json 2>/dev/null || exit $rc
  sleep 90
done
exit 3
