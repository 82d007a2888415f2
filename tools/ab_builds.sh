#!/bin/bash
# A/B of debug build variants on ONE box: tools/ab_builds.sh "<defs A>" "<defs B>" ...   ("-" = the shipped build)
# each: forced rebuild with MBX_BUILD_DEFS, then bench.py (train leg only, 30 steps); the shipped build is restored at the end.
cd "$(dirname "$0")/.."
for defs in "$@"; do
  d="$defs"; [ "$d" = "-" ] && d=""
  MBX_BUILD_DEFS="$d" python -c "from multibox_amd import build; build.build(force=True, verbose=False)" || exit 1
  for rep in 1 2; do
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-detect --no-roofline --no-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('defs [%s] rep $rep: %.3f ms/step %.1f img/s' % ('$defs', j['ms_per_step'], j['value']))"
  done
done
MBX_BUILD_DEFS="" python -c "from multibox_amd import build; build.build(force=True, verbose=False)"
