#!/bin/bash
# gpurun with a wait for a free slot: exit code 3 ("no box or slot free right now", nothing charged) is retried after a
# pause; any other outcome is returned as is.  usage: tools/gpurun_wait.sh <timeout> '<command>'
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
