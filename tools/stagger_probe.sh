#!/bin/bash
# start offsets between the workgroups of a persistent launch (MBX_STAGGER = quarter-microseconds per phase step, 4 phases;
# needs a DEBUG build: MBX_BUILD_DEFS=-DMBX_I5_STAMPS):
# the trunk-touching 1x1 launches (residual forward / accumulate + mask data gradient) at BATCH_SIZE 64 and 256
for b in ${BS:-64 256}; do for cfg in ${CFGS:-35 36 37 39}; do for st in ${STS:-0 4 8 16 32}; do
  echo -n "B=$b cfg=$cfg stagger=$st: "; for sh in ${SHAPES:-b17_up b35_up}; do KB_NO_WGRAD=1 KB_EPI=res KB_CFG=$cfg KB_B=$b KB_ONLY=$sh MBX_STAGGER=$st timeout -k 10 120 python tools/kbench.py 2>&1 | grep "^$sh" | awk '{printf "%s f %s d %s (%s %s) | ", $1, $2, $5, $10, $11}'; done; echo; done; done; done
