#!/bin/bash
# run the given steps one after another on the GPU box; a step that TIMES OUT or is killed ends the call (no further GPU
# step is started), an ordinary failure is recorded and the next step runs.  usage: gpu_steps.sh "name|seconds|command" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== $name (limit ${secs}s) $(date +%T)"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc $(date +%T)"
  tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping"; exit $rc; fi
done
exit 0
