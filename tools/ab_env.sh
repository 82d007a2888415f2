#!/bin/bash
# A/B of environment settings on one box: tools/ab_env.sh <reps> "<VAR=.. VAR=..>" "<VAR=..>" ...   (bench.py train leg only, 30
# steps; the settings alternate, "-" is the shipped default)
cd "$(dirname "$0")/.."
reps="$1"; shift
for rep in $(seq 1 "$reps"); do
  for setting in "$@"; do
    s="$setting"; [ "$s" = "-" ] && s=""
    env $s python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-detect --no-roofline --no-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %.3f ms/step %.1f img/s  kernels %s' % ('$setting', j['ms_per_step'], j['value'], j.get('kernels_per_step')))"
  done
done
