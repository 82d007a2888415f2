#!/bin/bash
# A/B of two BUILT libraries on one box: tools/ab_libs.sh <other libmbx.so> [reps] [VAR=val ...]   (alternates shipped /
# other; the shipped library is restored at the end; the VAR=val settings apply to the OTHER library's runs only -- e.g.
# MBX_RELU_BITS=0 for a library older than mbx_conv_desc.relu_bits).  bench.py train leg only, 30 steps.
cd "$(dirname "$0")/.."
other="$1"; reps="${2:-2}"; shift; shift; oenv="$*"
cp multibox_amd/libmbx.so /tmp/libmbx_shipped.so
run() {
  env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-detect --no-roofline --no-configs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: %.3f ms/step %.1f img/s' % (j['ms_per_step'], j['value']))"
}
for rep in $(seq 1 "$reps"); do
  cp /tmp/libmbx_shipped.so multibox_amd/libmbx.so && run shipped
  cp "$other" multibox_amd/libmbx.so && run other "$oenv"
done
cp /tmp/libmbx_shipped.so multibox_amd/libmbx.so
