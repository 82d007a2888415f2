"""TEST INFRASTRUCTURE ONLY -- the CPU baseline: one training step of the restated reference
(train.py:173-298) on torch-CPU float32: forward (oracle/torch_model.py), numpy/scipy matching
(oracle/ref_numpy.py, loss.py:8-53), loss, backward, RMSProp + EMA.  Used by bench.py's
`cpu_baseline` leg and by tests; never by the product path."""
from __future__ import annotations

import numpy as np
import torch

from . import ref_numpy as R
from .torch_model import Model, multibox_loss

WD = 0.00004


class CpuTrainer:
    def __init__(self, params, priors, k=5, fine_tune=True, alpha=1000.0, lr0=0.01, dsteps=7116, factor=0.94,
                 rms_decay=0.9, eps=1.0, ema_decay=0.9999):
        self.P = {n: v.clone() for n, v in params.items()}
        self.priors = torch.as_tensor(priors, dtype=torch.float32)
        self.k, self.fine_tune, self.alpha = k, fine_tune, alpha
        self.lr0, self.dsteps, self.factor, self.rms_decay, self.eps, self.ema_decay = lr0, dsteps, factor, rms_decay, eps, ema_decay
        train = lambda n: n.endswith(("/weights", "/biases", "/beta")) and (n.startswith("Multibox/") or not fine_tune)
        self.trainable = [n for n in self.P if train(n)]
        self.ms = {n: torch.ones_like(self.P[n]) for n in self.trainable}
        self.ema = {n: v.clone() for n, v in self.P.items()}
        self.step_no = 0

    def step(self, images, gt, n_gt):
        for n in self.trainable:
            self.P[n].requires_grad_(True)
            self.P[n].grad = None
        m = Model(self.P, k=self.k, bn_training=not self.fine_tune, heads_bn_training=True)
        x = images.permute(0, 3, 1, 2)
        if self.fine_tune:
            with torch.no_grad():
                feat = m.backbone(x)
        else:
            feat = m.backbone(x)
        locs, logits = m.heads(feat)
        dec = (locs.detach() + self.priors.unsqueeze(0)).numpy()
        conf = (torch.sigmoid(logits.detach()) + 1e-10).numpy().astype(np.float32)
        B = images.shape[0]
        _, _, match = R.compute_assignments(dec.reshape(-1, 4), conf.reshape(-1), gt, n_gt, B, self.alpha)
        loc, cf = multibox_loss(locs, logits, self.priors, torch.as_tensor(gt), match, self.alpha)
        reg = sum(0.5 * WD * (self.P[n] ** 2).sum() for n in self.P if n.endswith(("/weights", "/biases")))
        (loc + cf + reg).backward()
        lr = float(R.learning_rate(self.step_no, self.lr0, self.dsteps, self.factor))
        d = R.ema_decay(self.ema_decay, self.step_no)
        with torch.no_grad():
            for n, (mm, mv) in m.new_moving.items():
                self.P[n + "/BatchNorm/moving_mean"].copy_(mm)
                self.P[n + "/BatchNorm/moving_variance"].copy_(mv)
            for n in self.P:
                self.ema[n] -= (1 - d) * (self.ema[n] - self.P[n].detach())
            for n in self.trainable:
                g = self.P[n].grad
                self.ms[n].mul_(self.rms_decay).add_((1 - self.rms_decay) * g * g)
                self.P[n].detach().sub_(lr * g / torch.sqrt(self.ms[n] + self.eps))
        for n in self.trainable:
            self.P[n].requires_grad_(False)
        self.step_no += 1
        return float(loc), float(cf), float(reg)
