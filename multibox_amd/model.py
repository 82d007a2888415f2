"""Module-level mirror of the reference's model.build (model.py:326-337) on the libmbx engine.

    locs, confs, inception_vars = model.build(inputs, num_bboxes_per_cell, reuse=False, scope='')

`inputs` is a float32 CUDA tensor [B, S, S, 3] in [-1, 1] (inputs.py:350-351) instead of a TF tensor; the results are
the tensors the reference's graph nodes evaluate to: locations [B, P, 4] (raw residuals, model.py:320), confidences
[B, P, 1] after the sigmoid (model.py:322), and the dict {variable name: tensor} of the backbone's model variables
(model.py:332-333: weights, biases, BatchNorm beta / moving_mean / moving_variance under 'InceptionResnetV2/').

The reference picks the batch-norm mode with the arg_scope it wraps around the call (train.py:92-150,
detect.py:313-334); here it is the `mode` keyword: "infer" (default: every BN frozen, as detect.py builds it),
"train" (batch statistics) or "fine_tune" (backbone frozen, heads training).  The variables live in a Net that is
created on the first call for a (batch, size, k, mode) and reused when `reuse=True` -- tf.variable_scope reuse.
"""
from __future__ import annotations

import torch

from . import _lib
from .engine import Net

_NETS = {}


def get_net(batch, input_size, num_bboxes_per_cell, mode="infer"):
    return _NETS.get((batch, input_size, num_bboxes_per_cell, mode))


def build(inputs, num_bboxes_per_cell, reuse=False, scope="", mode="infer", seed=2):
    assert scope == "", "the reference always passes scope='' (train.py:111, detect.py:331)"
    assert mode in ("infer", "train", "fine_tune")
    assert inputs.dim() == 4 and inputs.shape[3] == 3 and inputs.shape[1] == inputs.shape[2], "inputs: [B, S, S, 3]"
    if not inputs.is_cuda:
        raise _lib.MbxError("model.build needs CUDA tensors: the network runs on libmbx only (no CPU fallback)")
    B, S = int(inputs.shape[0]), int(inputs.shape[1])
    key = (B, S, int(num_bboxes_per_cell), mode)
    net = _NETS.get(key)
    if net is None:
        if reuse:
            raise ValueError("reuse=True but no variables exist for %r (tf.variable_scope would raise too)" % (key,))
        net = Net(batch=B, input_size=S, k=int(num_bboxes_per_cell), mode="infer" if mode == "infer" else "train",
                  fine_tune=mode == "fine_tune", seed=seed)
        _NETS[key] = net
    elif not reuse:
        raise ValueError("variables for %r already exist; pass reuse=True (tf.variable_scope semantics)" % (key,))
    if mode != "train":
        net.fold_bn()                       # frozen BN layers read the CURRENT moving statistics
    net.set_input(inputs.to(torch.float32).contiguous())
    locs, logits = net.forward()
    confs = torch.empty_like(logits)
    _lib.check(_lib.lib().mbx_decode_conf(None, logits.data_ptr(), None, B, net.P, 0.0, None, confs.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "sigmoid")
    original_inception_vars = {n: net.get_param(n) for n in net.param_index if n.startswith("InceptionResnetV2/")}
    return locs, confs.reshape(B, net.P, 1), original_inception_vars
