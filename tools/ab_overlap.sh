set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_nnops.py -x -q -m gpu 2>&1 | tail -6
MBX_DETERMINISTIC=1 python tools/overlap_check.py 0 2 8 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -6
B="--steps 30 --warmup 5 --no-cpu-baseline --no-detect --no-roofline --no-configs"
for v in 1 0 1 0; do MBX_BN_GROUPS=$v python bench.py $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bn_groups=$v: %.3f ms/step %.1f img/s losses %s' % (j['ms_per_step'], j['value'], j['final_losses']))"; done
python tools/step_trace.py gpurun_out/step_trace_r4c.tsv 2>/dev/null | tail -30
