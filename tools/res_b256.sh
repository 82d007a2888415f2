#!/bin/bash
# resident-image launches at BATCH_SIZE 64 / 256 against the library's default tiles (kbench shapes), then the detect leg
for b in 64 256; do for c in 0 98; do echo "B=$b cfg=$c"; for sh in b17_1x7 b17_7x1 b8_1x3 b8_3x1; do KB_B=$b KB_CFG=$c KB_ONLY=$sh timeout -k 10 120 python tools/kbench.py 2>&1 | grep "^$sh" ; done; done; done
python tools/bench_detect.py 2>&1 | tail -1
