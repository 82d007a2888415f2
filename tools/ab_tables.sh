#!/bin/bash
# A/B of two tile tables on one box: tools/ab_tables.sh <other tune_cache.json> [reps]   (alternates shipped / other; the
# shipped table is restored at the end).  bench.py train leg only, 30 steps.
cd "$(dirname "$0")/.."
other="$1"; reps="${2:-2}"
cp multibox_amd/tune_cache.json /tmp/tune_shipped.json
run() {
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-detect --no-roofline --no-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.3f ms/step %.1f img/s  tune_cache %s' % (j['ms_per_step'], j['value'], j.get('tune_cache')))"
}
for rep in $(seq 1 "$reps"); do
  cp /tmp/tune_shipped.json multibox_amd/tune_cache.json && run shipped
  cp "$other" multibox_amd/tune_cache.json && run other
done
cp /tmp/tune_shipped.json multibox_amd/tune_cache.json
