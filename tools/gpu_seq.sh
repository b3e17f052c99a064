#!/bin/bash
# Run GPU steps one after the other on the GPU box; a step that TIMES OUT or is KILLED (124 / 137) ends the sequence (no further GPU step
# after a hung one), an ordinary failure does not.   usage: tools/gpu_seq.sh "<timeout s>|<log name>|<command>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  IFS='|' read -r to log cmd <<< "$spec"
  echo "=== [$log] $cmd"
  timeout -k 10 "$to" bash -c "$cmd" > "gpurun_out/$log" 2>&1
  rc=$?
  echo "=== [$log] rc=$rc"; tail -n 6 "gpurun_out/$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed: stopping the sequence"; exit $rc; fi
done
exit 0
