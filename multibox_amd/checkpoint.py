"""Checkpoint / resume: the reference saves every variable with tf.train.Saver
(train.py:282-286) and inference restores the EMA shadows into the live variables
(detect.py:336-346).  Here: one .pt file `model.ckpt-<global_step>.pt` holding the flat
buffers (weights, betas, moving statistics, RMSProp slots, EMA shadows, step) plus the
name index; TensorFlow V1 checkpoints are read by multibox_amd/tf_checkpoint.py into the same structure."""
import glob
import os
import re

import torch


def _prune(logdir, max_to_keep, keep_every_n_hours, now=None):
    """tf.train.Saver's retention (train.py:282-286): the newest max_to_keep checkpoints stay; an older one that is about to
    be deleted is KEPT FOR GOOD instead if it was written at least keep_checkpoint_every_n_hours after the previous
    permanently kept one (Saver._next_checkpoint_time; the clock starts at the first save into this directory).  The
    little state TF keeps in the Saver object lives in <logdir>/checkpoint_retention.json here, so that it survives restarts
    (TF's own clock restarts with every process: a run resumed often keeps a checkpoint later than tf.train.Saver would).
    A state file that does not hold what is expected (hand-edited, an older format, `null`) is re-initialised."""
    import json
    import time
    state_path = os.path.join(logdir, "checkpoint_retention.json")
    st = None
    try:
        with open(state_path) as f:
            st = json.load(f)
    except (OSError, ValueError):
        pass
    if not (isinstance(st, dict) and isinstance(st.get("kept"), list) and isinstance(st.get("next_keep_time"), (int, float))
            and all(isinstance(k, str) for k in st["kept"])):
        st = {"next_keep_time": (now if now is not None else time.time()) + 3600.0 * keep_every_n_hours, "kept": []}
    kept = set(st["kept"])
    olds = sorted((p for p in glob.glob(os.path.join(logdir, "model.ckpt-*.pt")) if os.path.basename(p) not in kept), key=_step_of)
    for p in olds[:-max_to_keep] if max_to_keep > 0 else []:
        t = os.path.getmtime(p)
        if keep_every_n_hours > 0 and t > st["next_keep_time"]:
            st["kept"].append(os.path.basename(p))
            st["next_keep_time"] += 3600.0 * keep_every_n_hours
        else:
            os.remove(p)
    tmp = state_path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(st, f)
    os.replace(tmp, state_path)


def save(logdir, trainer, max_to_keep=3, keep_checkpoint_every_n_hours=10000.0):
    net = trainer.net
    os.makedirs(logdir, exist_ok=True)
    # (only APPLIED steps count: steps skipped since the trainer's last health check are taken off here without touching its state)
    gstep = trainer.applied_global_step() if hasattr(trainer, "applied_global_step") else trainer.global_step
    path = os.path.join(logdir, "model.ckpt-%d.pt" % gstep)
    state = dict(global_step=gstep, k=net.k, input_size=net.S,
                 index={n: (b, o, tuple(s), c) for n, (b, o, s, c) in net.param_index.items()},
                 W=net.W.cpu(), Bt=net.Bt.cpu(), MM=net.MM.cpu(), MV=net.MV.cpu(),
                 Wms=trainer.Wms.cpu(), Btms=trainer.Btms.cpu(),
                 Wema=trainer.Wema.cpu(), Btema=trainer.Btema.cpu(), MMema=trainer.MMema.cpu(), MVema=trainer.MVema.cpu())
    if trainer.Wmom is not None:                    # RMSPROP_MOMENTUM != 0: the momentum slots are optimiser state too
        state.update(Wmom=trainer.Wmom.cpu(), Btmom=trainer.Btmom.cpu())
    tmp = path + ".tmp"                             # a kill in mid-save must not leave a truncated newest checkpoint
    torch.save(state, tmp)
    os.replace(tmp, path)
    try:                                            # a retention problem must never lose the checkpoint just written
        _prune(logdir, max_to_keep, float(keep_checkpoint_every_n_hours))
    except Exception as e:                          # noqa: BLE001
        import sys
        print("[multibox_amd] checkpoint retention skipped: %r" % (e,), file=sys.stderr)
    return path


def _step_of(path):
    m = re.search(r"-(\d+)\.pt$", path)
    return int(m.group(1)) if m else -1


def latest_checkpoint(path):
    """tf.train.latest_checkpoint analogue: a file, or the newest model.ckpt-*.pt of a directory -- or, in a directory
    written by TensorFlow, the file its `checkpoint` state file names (model_checkpoint_path: "...")."""
    if os.path.isdir(path):
        c = sorted(glob.glob(os.path.join(path, "model.ckpt-*.pt")), key=_step_of)
        if c:
            return c[-1]
        state = os.path.join(path, "checkpoint")
        if os.path.exists(state):
            m = re.search(r'^model_checkpoint_path:\s*"([^"]+)"', open(state).read(), re.M)
            if m:
                cand = m.group(1) if os.path.isabs(m.group(1)) else os.path.join(path, m.group(1))
                return cand if os.path.exists(cand) else None
        return None
    return path if os.path.exists(path) else None


def is_tf_checkpoint(path):
    """A TensorFlow V1 checkpoint table (tf.train.Saver of TF <= 0.11): recognised by the table magic."""
    from . import tf_checkpoint as TF
    try:
        with open(path, "rb") as f:
            f.seek(-8, os.SEEK_END)
            return int.from_bytes(f.read(8), "little") == TF.MAGIC
    except OSError:
        return False


def global_step_of(path):
    """detect.py:383: the global step is parsed from the checkpoint file name."""
    m = re.search(r"-(\d+)(\.pt)?$", path)
    return int(m.group(1)) if m else -1


def restore_pretrained(path, trainer, fine_tune=False, use_moving_averages=False, restore_moving_averages=False):
    """train.py:15-90 (get_init_function): initialise from --pretrained_model.  A TensorFlow checkpoint goes through
    multibox_amd.tf_checkpoint with the reference's three switches; one of this build's own .pt files restores
    everything it holds (variables, slots, shadows)."""
    ck = latest_checkpoint(path)
    if ck is None:
        raise FileNotFoundError("no checkpoint found at %s" % path)
    if is_tf_checkpoint(ck):
        from . import tf_checkpoint as TF
        net = trainer.net
        TF.restore(ck, net, fine_tune=fine_tune, use_moving_averages=use_moving_averages,
                   restore_moving_averages=restore_moving_averages, ema=trainer)
        net.refresh_bf16()
        if net.fine_tune:
            net.fold_bn()
    else:
        restore_for_training(ck, trainer)
    trainer.global_step = 0
    trainer.refresh_frozen_reg()
    return ck


def restore_for_training(path, trainer):
    st = torch.load(path, map_location="cpu")
    net = trainer.net
    assert st["W"].numel() == net.nW and st["k"] == net.k, "checkpoint does not match the network"
    for name, t in (("W", net.W), ("Bt", net.Bt), ("MM", net.MM), ("MV", net.MV)):
        t.copy_(st[name])
    for name in ("Wms", "Btms", "Wema", "Btema", "MMema", "MVema"):
        getattr(trainer, name).copy_(st[name])
    if trainer.Wmom is not None and "Wmom" in st:
        trainer.Wmom.copy_(st["Wmom"])
        trainer.Btmom.copy_(st["Btmom"])
    trainer.global_step = int(st["global_step"])
    net.refresh_bf16()
    if net.fine_tune:
        net.fold_bn()
    trainer.refresh_frozen_reg()


def restore_for_inference(path, net, use_moving_averages=True):
    """detect.py:336-346: live variables <- EMA shadows."""
    if is_tf_checkpoint(path):
        from . import tf_checkpoint as TF
        step = TF.restore_for_inference(path, net, use_moving_averages=use_moving_averages)
        net.Wb.copy_(net.W.to(torch.bfloat16))
        net.fold_bn()
        return step if step else max(global_step_of(path), 0)
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
