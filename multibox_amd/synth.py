"""Synthetic inputs of the benchmark / tests: the reference has no generator, so the
shapes follow its input contract (inputs.py:340-351): images [B,S,S,3] float32 in [-1,1], ground
truth [B,G,4] zero-padded x1,y1,x2,y2 in [0,1], counts [B] int32."""
import numpy as np


def synthetic_batch(batch, size=299, max_num_bboxes=13, seed=0):
    rng = np.random.RandomState(seed)
    images = rng.uniform(-1.0, 1.0, (batch, size, size, 3)).astype(np.float32)
    rng = np.random.RandomState(seed + 1)
    n = rng.randint(0, max_num_bboxes + 1, batch).astype(np.int32)      # includes images without boxes
    gt = np.zeros((batch, max_num_bboxes, 4), np.float32)
    for b in range(batch):
        xy = rng.uniform(0, 0.7, (n[b], 2))
        wh = rng.uniform(0.05, 0.3, (n[b], 2))
        gt[b, :n[b], :2] = xy
        gt[b, :n[b], 2:] = xy + wh
    return images, gt, n


DEFAULT_ASPECT_RATIOS = {5: [1.0, 2.0, 3.0, 1.0 / 2.0, 1.0 / 3.0],
                         7: [1.0, 2.0, 3.0, 1.0 / 2.0, 1.0 / 3.0, 1.5, 1.0 / 1.5]}
