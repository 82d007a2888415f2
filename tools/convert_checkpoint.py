"""Write one of this build's checkpoints (model.ckpt-N.pt) as a TensorFlow V1 checkpoint table with slim's variable
names (HWIO filters, `<var>/ExponentialMovingAverage` shadows, global_step), the layout tf.train.Saver of TF <= 0.11
reads and the reference's train.py:15-90 / detect.py:336-346 restore from.  The opposite direction needs no tool:
train.py --pretrained_model, detect.py / eval.py --checkpoint_path take a TensorFlow checkpoint directly.
usage: python tools/convert_checkpoint.py LOGDIR_OR_PT OUT_CKPT"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    import torch
    from multibox_amd import checkpoint as CK, tf_checkpoint as TF
    from multibox_amd.engine import Net
    src = CK.latest_checkpoint(sys.argv[1])
    if src is None or not src.endswith(".pt"):
        raise SystemExit("no model.ckpt-*.pt at %s" % sys.argv[1])
    st = torch.load(src, map_location="cpu")
    net = Net(batch=1, input_size=st["input_size"], k=st["k"], mode="train", device="cpu")
    assert st["W"].numel() == net.nW, "checkpoint does not match the network"
    for name, t in (("W", net.W), ("Bt", net.Bt), ("MM", net.MM), ("MV", net.MV)):
        t.copy_(st[name])
    ema = types.SimpleNamespace(Wema=st["Wema"], Btema=st["Btema"], MMema=st["MMema"], MVema=st["MVema"])
    TF.export(sys.argv[2], net, ema=ema, global_step=int(st["global_step"]))
    print("wrote %s (%d variables + shadows, global_step %d)" % (sys.argv[2], len(net.param_index), int(st["global_step"])))


if __name__ == "__main__":
    main()
