#!/bin/bash
# usage: tools/hb.sh <command...> -- runs the command with a heartbeat file under gpurun_out/ (gpurun kills a run that writes
# nothing for 7 minutes; a long CPU-oracle test is silent for longer), unbuffered Python output.
mkdir -p gpurun_out
( while true; do date +%s > gpurun_out/.heartbeat; sleep 45; done ) &
HB=$!
export PYTHONUNBUFFERED=1
"$@"
rc=$?
kill $HB 2>/dev/null
exit $rc
