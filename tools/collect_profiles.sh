#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun; outputs under gpurun_out/prof_$1).
#   part a: the default bench line, kernel stats + per-layer table of the traced bench, FETCH_SIZE / WRITE_SIZE passes
#           (separate --pmc runs, eager steps; tools/pmc_traffic.py applies the gfx950 FETCH_SIZE x2 correction)
#   part b: SQ counters of every convolution kernel instantiation of the step (two --pmc passes), the grouped weight gradient
#   part c: the 2000-step saturation stress, the 512x512 configuration
#   part d: the detect leg: per-layer table + FETCH_SIZE / WRITE_SIZE passes (profiles/rNN_detect_traffic_pmc.json)
# Counters never share a run with other trace domains than --kernel-trace.
set -o pipefail
tag=${1:-r04}
part=${2:-a}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline --no-detect --no-configs"
if [ "$part" = "a" ]; then
  python bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-detect --no-roofline --no-configs > $out/bench_prof.json 2> $out/bench_prof.err || exit 2
  python tools/trace_by_layer.py $out/kt/kt_kernel_trace.csv > $out/by_layer.txt 2>&1
  cp $out/kt/kt_kernel_stats.csv $out/kernel_stats.csv
  python tools/step_trace.py --csv $out/kt/kt_kernel_trace.csv $out/step_trace.tsv > $out/step_trace.txt 2>&1
  rm -rf $out/kt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -o f -- python3 bench.py $B > $out/pmc_f.log 2>&1 || exit 3
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -o w -- python3 bench.py $B > $out/pmc_w.log 2>&1 || exit 4
  python tools/pmc_traffic.py $out/pf/f_counter_collection.csv $out/pw/w_counter_collection.csv 3 $out/hbm_traffic_pmc.json > $out/pmc_traffic.txt 2>&1
  rm -rf $out/pf $out/pw
elif [ "$part" = "t" ]; then
  # the step trace alone (every kernel of the last complete step of a traced bench run, in launch order)
  rocprofv3 --kernel-trace --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-detect --no-roofline --no-configs > $out/bench_trace.json 2> $out/bench_trace.err || exit 2
  python tools/step_trace.py --csv $out/kt/kt_kernel_trace.csv $out/step_trace.tsv > $out/step_trace.txt 2>&1
  rm -rf $out/kt
  cat $out/step_trace.txt
elif [ "$part" = "b" ]; then
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $out/sq1 -o s -- python3 bench.py $B > $out/sq1.log 2>&1 || exit 5
  python tools/pmc_summary.py $out/sq1/s_counter_collection.csv > $out/conv_sq_counters.txt 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/sq2 -o s -- python3 bench.py $B > $out/sq2.log 2>&1 || exit 6
  python tools/pmc_summary.py $out/sq2/s_counter_collection.csv > $out/conv_lds_counters.txt 2>&1
  rm -rf $out/sq1 $out/sq2
elif [ "$part" = "w" ]; then
  # the grouped weight-gradient launches of the step: SQ counters, then L2 (TCC) counters, eager steps of the real network
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $out/wg1 -o s -- python3 bench.py $B > $out/wg1.log 2>&1 || exit 8
  python tools/pmc_summary.py $out/wg1/s_counter_collection.csv | grep -A 9 "conv_wgrad_grouped" > $out/wgrad_sq_counters.txt 2>&1
  # (the counter set of r03_conv_l2_counters.txt; under its own timeout: a TCC pass has hung inside rocprofv3 on this pool before)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr --output-format csv -d $out/wg2 -o s -- python3 bench.py $B > $out/wg2.log 2>&1 || exit 9
  python tools/pmc_summary.py $out/wg2/s_counter_collection.csv | grep -A 6 "conv_wgrad_grouped" > $out/wgrad_l2_counters.txt 2>&1
  rm -rf $out/wg1 $out/wg2
elif [ "$part" = "d" ]; then
  # the detect leg (BASELINE config 4): per-layer table, FETCH_SIZE / WRITE_SIZE passes of the detect forward
  python tools/detect_by_layer.py > $out/detect_by_layer.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/df -o f -- python3 tools/bench_detect.py > $out/dpmc_f.log 2>&1 || exit 10
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/dw -o w -- python3 tools/bench_detect.py > $out/dpmc_w.log 2>&1 || exit 11
  python tools/pmc_traffic.py $out/df/f_counter_collection.csv $out/dw/w_counter_collection.csv 15 $out/detect_traffic_pmc.json > $out/detect_pmc_traffic.txt 2>&1
  rm -rf $out/df $out/dw
elif [ "$part" = "c" ]; then
  MBX_DETERMINISTIC=1 python tools/side_stream_stress.py 2000 compare saturate > $out/saturation_stress.json 2> >(tee $out/saturation_stress.err >&2) || exit 7
  python bench.py --input-size 512 --k 7 --max-num-bboxes 100 --no-cpu-baseline --no-detect --no-configs > $out/bench_512.json 2> $out/bench_512.err
fi
ls -la $out
