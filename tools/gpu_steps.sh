#!/bin/bash
# Runs the steps of one gpurun call in order: tools/gpu_steps.sh <outdir> "<seconds>|<name>|<command>" ...
# A step that fails in the ordinary way (a test that does not pass, a script error) is logged and the next one runs; a step that is
# killed at its time limit (exit code 124 / 137) ends the call: no further GPU step is started after a kill.
O=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  secs="${spec%%|*}"; rest="${spec#*|}"; name="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== $name (limit $secs s): $cmd"
  t0=$(date +%s)
  timeout -k 10 $secs bash -c "$cmd" > $O/$name.log 2>&1
  rc=$?
  echo "=== $name: exit code $rc after $(( $(date +%s) - t0 )) s"
  tail -n 6 $O/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name was killed at its limit: stopping here"; exit $rc; fi
done
exit 0
