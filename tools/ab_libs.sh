#!/bin/bash
# A/B of two BUILT libraries on one box: tools/ab_libs.sh <other libmbx.so> [reps]   (alternates shipped / other; the
# shipped library is restored at the end).  bench.py train leg only, 30 steps.
cd "$(dirname "$0")/.."
other="$1"; reps="${2:-2}"
cp multibox_amd/libmbx.so /tmp/libmbx_shipped.so
run() {
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-detect --no-roofline --no-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.3f ms/step %.1f img/s' % (j['ms_per_step'], j['value']))"
}
for rep in $(seq 1 "$reps"); do
  cp /tmp/libmbx_shipped.so multibox_amd/libmbx.so && run shipped
  cp "$other" multibox_amd/libmbx.so && run other
done
cp /tmp/libmbx_shipped.so multibox_amd/libmbx.so
