set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_nnops.py -x -q -m gpu 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py tests/test_gpu_assembled.py -x -q -m gpu -k "not saturation and not side_stream and not soak" 2>&1 | tail -6
bash tools/ab_env.sh - "MBX_POOL_FUSE=0" | grep rep
python tools/step_trace.py gpurun_out/step_trace_r4f.tsv 2>/dev/null | tail -31
