#!/bin/bash
# Queueing helper (runs HERE, not on the GPU box): calls gpurun and, ONLY when it answers "no box or slot free right now"
# (exit code 3: nothing ran, nothing was charged), waits and asks again.  Any other outcome -- success, a failing command, a
# refusal -- is returned at once: a GPU step that ran is never run again by this script.
# usage: tools/gpurun_wait.sh TIMEOUT_SECONDS 'command'
T=$1; shift
for try in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 75
done
exit 3
