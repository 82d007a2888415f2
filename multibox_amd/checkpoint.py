"""Checkpoint / resume: the reference saves every variable with tf.train.Saver
(train.py:282-286) and inference restores the EMA shadows into the live variables
(detect.py:336-346).  Here: one .pt file `model.ckpt-<global_step>.pt` holding the flat
buffers (weights, betas, moving statistics, RMSProp slots, EMA shadows, step) plus the
name index, so a TF checkpoint importer can fill the same structure later."""
import glob
import os
import re

import torch


def save(logdir, trainer, max_to_keep=3):
    net = trainer.net
    os.makedirs(logdir, exist_ok=True)
    path = os.path.join(logdir, "model.ckpt-%d.pt" % trainer.global_step)
    state = dict(global_step=trainer.global_step, k=net.k, input_size=net.S,
                 index={n: (b, o, tuple(s), c) for n, (b, o, s, c) in net.param_index.items()},
                 W=net.W.cpu(), Bt=net.Bt.cpu(), MM=net.MM.cpu(), MV=net.MV.cpu(),
                 Wms=trainer.Wms.cpu(), Btms=trainer.Btms.cpu(),
                 Wema=trainer.Wema.cpu(), Btema=trainer.Btema.cpu(), MMema=trainer.MMema.cpu(), MVema=trainer.MVema.cpu())
    torch.save(state, path)
    olds = sorted(glob.glob(os.path.join(logdir, "model.ckpt-*.pt")), key=_step_of)
    for p in olds[:-max_to_keep]:
        os.remove(p)
    return path


def _step_of(path):
    m = re.search(r"-(\d+)\.pt$", path)
    return int(m.group(1)) if m else -1


def latest_checkpoint(path):
    """tf.train.latest_checkpoint analogue: a file, or the newest model.ckpt-*.pt of a directory."""
    if os.path.isdir(path):
        c = sorted(glob.glob(os.path.join(path, "model.ckpt-*.pt")), key=_step_of)
        return c[-1] if c else None
    return path if os.path.exists(path) else None


def global_step_of(path):
    """detect.py:383: the global step is parsed from the checkpoint file name."""
    return _step_of(path)


def restore_for_training(path, trainer):
    st = torch.load(path, map_location="cpu")
    net = trainer.net
    assert st["W"].numel() == net.nW and st["k"] == net.k, "checkpoint does not match the network"
    for name, t in (("W", net.W), ("Bt", net.Bt), ("MM", net.MM), ("MV", net.MV)):
        t.copy_(st[name])
    for name in ("Wms", "Btms", "Wema", "Btema", "MMema", "MVema"):
        getattr(trainer, name).copy_(st[name])
    trainer.global_step = int(st["global_step"])
    net.refresh_bf16()
    if net.fine_tune:
        net.fold_bn()


def restore_for_inference(path, net, use_moving_averages=True):
    """detect.py:336-346: live variables <- EMA shadows."""
    st = torch.load(path, map_location="cpu")
    assert st["W"].numel() == net.nW and st["k"] == net.k, "checkpoint does not match the network"
    sfx = "ema" if use_moving_averages else ""
    net.W.copy_(st["W" + sfx])
    net.Bt.copy_(st["Bt" + sfx])
    net.MM.copy_(st["MM" + sfx])
    net.MV.copy_(st["MV" + sfx])
    net.Wb.copy_(net.W.to(torch.bfloat16))
    net.fold_bn()
    return int(st["global_step"])
