set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_assembled.py tests/test_gpu_conv.py tests/test_gpu_dp.py tests/test_gpu_nnops.py -x -q -m gpu 2>&1 | tail -15
python bench.py > gpurun_out/bench_new.json 2> gpurun_out/bench_new.err; echo "bench rc $?"; tail -c 6000 gpurun_out/bench_new.json
