set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "split_k" 2>&1 | tail -4
B="--steps 30 --warmup 5 --no-cpu-baseline --no-detect --no-roofline --no-configs"
for v in 1 0 1 0; do MBX_SPLITK=$v python bench.py $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('splitk=$v: %.3f ms/step %.1f img/s' % (j['ms_per_step'], j['value']))"; done
python tools/step_trace.py gpurun_out/step_trace_r4b.tsv 2>/dev/null | grep -i "splitk\|step wall"
grep -n "splitk_reduce" -B1 gpurun_out/step_trace_r4b.tsv
