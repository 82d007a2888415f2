set -o pipefail
cd $GRAFT_REPO_ROOT
MBX_WG_OVERLAP=0 python tools/overlap_trace.py 2>/dev/null
MBX_WG_OVERLAP=96 MBX_WG_GROUPS=8 python tools/overlap_trace.py 2>/dev/null
MBX_WG_OVERLAP=32 MBX_WG_GROUPS=8 python tools/overlap_trace.py 2>/dev/null
