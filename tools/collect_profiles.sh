#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun; outputs under gpurun_out/prof_$1):
#   kernel stats of the default bench, FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, eager steps), per-layer table.
set -o pipefail
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-detect --no-roofline > $out/bench_prof.json 2> $out/bench_prof.err || exit 2
python tools/trace_by_layer.py $out/kt/kt_kernel_trace.csv > $out/by_layer.txt 2>&1
cp $out/kt/kt_kernel_stats.csv $out/kernel_stats.csv
rm -f $out/kt/kt_kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -o f -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline --no-detect > $out/pmc_f.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -o w -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline --no-detect > $out/pmc_w.log 2>&1 || exit 4
python tools/pmc_traffic.py $out/pf/f_counter_collection.csv $out/pw/w_counter_collection.csv 3 $out/hbm_traffic_pmc.json > $out/pmc_traffic.txt 2>&1
rm -rf $out/pf $out/pw $out/kt
ls -la $out
