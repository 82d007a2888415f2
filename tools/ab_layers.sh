#!/bin/bash
# per-layer convolution times of two environment settings on ONE box (rocprofv3 kernel trace of a short bench run each):
# tools/ab_layers.sh "<VAR=val ...>" "<...>"   ("-" = defaults) -> gpurun_out/ab_layers_<i>.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
i=0
for envs in "$@"; do
  e="$envs"; [ "$e" = "-" ] && e=""
  rm -rf /tmp/abl_$i
  for kv in $e; do export "$kv"; done
  rocprofv3 --kernel-trace -d /tmp/abl_$i -o x --output-format csv -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-detect --no-roofline --no-configs > /dev/null 2>&1
  for kv in $e; do unset "${kv%%=*}"; done
  f=$(find /tmp/abl_$i -name "x_kernel_trace.csv" | head -1)
  python tools/trace_by_layer.py "$f" > gpurun_out/ab_layers_$i.txt 2>&1
  echo "[$envs] -> gpurun_out/ab_layers_$i.txt: $(grep '^totals' gpurun_out/ab_layers_$i.txt | head -1)"
  i=$((i+1))
done
