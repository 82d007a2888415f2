#!/bin/bash
# head lanes (MBX_HEAD_LANES=1) + the clearing launch beside the forward pass (MBX_ZERO_BESIDE=1), both off by default:
# same-box A/B of the training step, the fine-tune leg and the detect leg
for rep in 1 2; do for v in "1 1" "0 0" "1 0" "0 1"; do set -- $v; echo -n "lanes=$1 zero_beside=$2: "; MBX_HEAD_LANES=$1 MBX_ZERO_BESIDE=$2 python bench.py --no-cpu-baseline --no-detect --no-roofline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', j['ms_per_step'], 'fine_tune', j['configs']['fine_tune']['ms_per_step'], 'losses', j['final_losses'])"; done; done
for l in 1 0; do echo -n "lanes=$l detect: "; MBX_HEAD_LANES=$l python tools/bench_detect.py 2>&1 | tail -1 | cut -c1-110; done
