"""Detection path measurement (BASELINE config 4): forward (inference BN, bf16) + decode/filter/top-K at
BATCH_SIZE=256 patches, k=7 -> P=904, whole-image restrictions, max_to_keep 200.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
from multibox_amd.engine import Net
from multibox_amd import priors as PR, detect as D, _lib
from multibox_amd.synth import DEFAULT_ASPECT_RATIOS

B, k, S = int(os.environ.get("DB_B", "256")), 7, 299
priors = np.array(PR.generate_priors(DEFAULT_ASPECT_RATIOS[k]), np.float32)
net = Net(batch=B, input_size=S, k=k, mode="infer")
net.fold_bn()
images = torch.rand(B, S, S, 3, device="cuda") * 2 - 1
meta = D.make_patch_meta(np.zeros((B, 2), np.int32), np.tile([[S, S]], (B, 1)), np.zeros((B, 1), np.int32),
                         np.tile([[0., 0., 1., 1.]], (B, 1)), np.full((B, 1), 200), np.tile([[S, S]], (B, 1)))
pp = D.DetectPostprocess(priors, B, k_max=200)
conf = torch.empty((B, net.P), device="cuda")
l = _lib.lib()

def fwd():
    net.set_input(images)
    net.forward()
    _lib.check(l.mbx_decode_conf(None, net.logits.data_ptr(), None, B, net.P, 0.0, None, conf.data_ptr(), torch.cuda.current_stream().cuda_stream))

def post():
    pp(net.locs, conf, meta)

for _ in range(2):
    fwd(); post()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    fwd(); post()
for _ in range(3):
    gr.replay()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    gr.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    post()
b.record()
torch.cuda.synchronize()
post_us = a.elapsed_time(b) / 20 * 1e3
bytes_alg = B * (20 * net.P + 24 * 200)
print(json.dumps({"metric": "images/sec detect (299x299 patches, k=7, P=%d)" % net.P, "value": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
                  "batch": B, "postprocess_us": round(post_us, 1), "postprocess_algorithmic_GBps": round(bytes_alg / post_us / 1e3, 2),
                  "forward_tflops": round(26.8e-3 * B / dt, 1)}))
