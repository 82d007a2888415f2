# What bounds the K loop of the resident-image launch: kbench of block17's 1x7 / 7x1 on debug builds that leave out the fragment
# reads (MBX_RES_PROBE=1), the MFMAs (2) or both (3) at COMPILE time (wrong results; the schedule of a step is otherwise unchanged).
# usage (through gpurun): bash tools/res_probe.sh
for pr in 0 1 2 3; do
  MBX_BUILD_DEFS=-DMBX_RES_PROBE=$pr python -c "
from multibox_amd import build as B; B.build(force=True, verbose=False)"
  echo "MBX_RES_PROBE=$pr"
  KB_ONLY=b17_ KB_NO_WGRAD=1 KB_CFG=98 timeout -k 10 200 python tools/kbench.py 2>&1 | grep -E "1x7|7x1"
done
python -c "
from multibox_amd import build as B; B.build(force=True, verbose=False)"
