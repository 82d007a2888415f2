#!/bin/bash
# row-wise BN kernels walking their rows by XCD-logical id (MBX_XCD_ROWS=1; off by default) against the grid-strided / round-robin walk (0):
# same-box A/B of the training step (+ fine-tune and 512 legs), alternating
for rep in 1 2 3; do for v in 1 0; do echo -n "xcd_rows=$v: "; MBX_XCD_ROWS=$v python bench.py --no-cpu-baseline --no-detect --no-roofline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', j['ms_per_step'], 'fine_tune', j['configs']['fine_tune']['ms_per_step'], '512:', j['configs']['s512_k7_g100']['ms_per_step'])"; done; done
