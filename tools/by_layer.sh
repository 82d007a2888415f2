set -o pipefail
out=gpurun_out/prof_tmp
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-detect --no-roofline --no-configs > $out/bench_prof.json 2> $out/bench_prof.err || exit 2
python tools/trace_by_layer.py $out/kt/kt_kernel_trace.csv > $out/by_layer.txt 2>&1
cp $out/kt/kt_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/kt
grep -E "block17|totals|conv_resident|igemm5_kernel  |bn_" $out/by_layer.txt
