set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_nnops.py -x -q -m gpu 2>&1 | tail -6
for v in 0 1; do MBX_POOL_FUSE=$v MBX_DETERMINISTIC=1 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd.trainer import Trainer, decay_steps
from multibox_amd import priors as PR
from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS
B = 64
priors = PR.priors_for_input_size(DEFAULT_ASPECT_RATIOS[5], 299).astype(np.float32)
images, gt, n = synthetic_batch(B, 299, 13, seed=0)
net = Net(batch=B, input_size=299, k=5, mode="train", seed=2, repeats=(1, 1, 1))
tr = Trainer(net, priors, max_num_bboxes=13, location_loss_alpha=1000.0, decay_steps_=decay_steps(56945, B, 4), use_graph=True)
tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
for _ in range(2):
    tr.step()
torch.cuda.synchronize()
import hashlib
print("POOL_FUSE", os.environ["MBX_POOL_FUSE"], "fused:", sum(1 for c in net.convs if getattr(c, "fused_pool", None) is not None),
      hashlib.sha256(net.W.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha256(net.Wg.cpu().numpy().tobytes()).hexdigest()[:16], tr.losses())
PY
done
B="--steps 30 --warmup 5 --no-cpu-baseline --no-detect --no-roofline --no-configs"
for v in 1 0 1 0; do MBX_POOL_FUSE=$v python bench.py $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pool_fuse=$v: %.3f ms/step %.1f img/s' % (j['ms_per_step'], j['value']))"; done
python tools/step_trace.py gpurun_out/step_trace_r4e.tsv 2>/dev/null | tail -31
