#!/bin/bash
# Runs GPU steps one after another on the gpurun box; a step that times out (rc 124 / 137) ends the call: no further GPU step
# is started after a kill.  usage: tools/gpu_steps.sh "<timeout s>|<log name>|<command>" ...
export TMPDIR=/tmp
mkdir -p gpurun_out
for spec in "$@"; do
    t="${spec%%|*}"; rest="${spec#*|}"; log="${rest%%|*}"; cmd="${rest#*|}"
    echo "=== [$t s] $cmd  -> gpurun_out/$log"
    timeout -k 10 "$t" bash -c "$cmd" > "gpurun_out/$log" 2>&1
    rc=$?
    echo "rc $rc" >> "gpurun_out/$log"
    tail -n 12 "gpurun_out/$log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: stopping"; exit $rc; fi
done
exit 0
