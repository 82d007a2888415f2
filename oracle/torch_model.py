"""TEST INFRASTRUCTURE ONLY -- torch-CPU float32 restatement of the reference network.

Follows /root/reference/model.py:6-337 layer by layer (block35/17/8, inception_resnet_v2,
build_detection_heads) with the slim/TF-0.11 semantics the reference relies on but does not
vendor: 'SAME' padding puts the odd pixel at the bottom/right, avg_pool SAME divides by the
number of valid taps, batch_norm has beta only (scale=False), epsilon 0.001, biased batch
variance (train.py:94-99).  The GRAPH is pinned: tests/golden/model_graph.json is generated from the reference's
model.py by tools/gen_model_graph.py (AST walk) and tests/test_model_graph.py checks this model's execution trace
against it layer by layer (scope, channels, kernel, stride, padding, BN / bias / activation, residual scale,
input tensors, head flatten order).  The NUMERICS of conv / BN / pool are TF-0.11's (un-vendored, cannot run
here): "parity unpinned" against TF itself for those.

Parameters come in a dict keyed by the slim variable names (".../weights" KRSC float32,
".../biases", ".../BatchNorm/beta|moving_mean|moving_variance").  ``q`` optionally rounds
tensors where the HIP engine stores bf16, so the two can be compared tightly.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 0.001


def q_bf16(t):
    """Round to bf16 with a straight-through gradient."""
    return t + (t.to(torch.bfloat16).to(torch.float32) - t).detach()


def _tag(t, prov):
    """Provenance for the layer-table parity test: which layer scopes produced the channels of t, in order."""
    t._prov = list(prov)
    return t


def _prov(t):
    return list(getattr(t, "_prov", ["?"]))


def _same(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


class Model:
    def __init__(self, params, k=5, bn_training=True, heads_bn_training=None, q=None, bn_decay=0.9997,
                 repeats=(10, 20, 9), force=None):
        """force: optional {scope: tensor [B,C,H,W]} of activations to TEACHER-FORCE: the output of every batch-norm
        convolution named in it (after BN / ReLU) and of every residual block named by its ".../Conv2d_1x1" scope takes
        the given VALUE while gradients still flow through this model's own graph (x + (forced - x).detach()).  With
        the engine's stored activations forced in, both backward passes are linearised around the same point (same
        ReLU masks, same layer inputs), which is what makes a full-depth gradient comparison well-posed: a bf16
        rounding difference in the forward pass otherwise flips ReLU masks and decorrelates the gradients of a random-init
        100-layer network (the bf16-emulating and the float32 oracle agree only to cosine ~0.5 there)."""
        self.force = force
        self.keep_acts, self.acts = False, {}          # debugging: keep every BN convolution's output (and its gradient)
        self.conv_io = {}                              # ... and (input, pre-BN output with .grad) of every convolution
        self.P, self.k, self.q = params, k, (q or (lambda t: t))
        self.repeats = repeats              # model.py:142,162,187 use (10, 20, 9); smaller = reduced-depth test net
        self.bn_training = bn_training
        self.heads_bn_training = bn_training if heads_bn_training is None else heads_bn_training
        self.bn_decay = bn_decay
        self.new_moving = {}
        self.endpoints = {}
        self._in_heads = False
        self.trace = []                     # one entry per conv / pool executed, in order (tests/test_model_graph.py)

    # slim.conv2d (+ batch_norm + relu)
    def conv(self, x, scope, ksize, stride=1, padding="SAME", bn=True, relu=True, bias=False):
        w = self.P[scope + "/weights"]                       # [K,R,S,C]
        kh, kw = (ksize, ksize) if isinstance(ksize, int) else ksize
        assert w.shape[1:3] == (kh, kw), (scope, w.shape)
        entry = {"scope": scope, "op": "conv2d", "in_channels": int(x.shape[1]), "out_channels": int(w.shape[0]),
                 "kernel": [kh, kw], "stride": stride, "padding": padding, "inputs": _prov(x), "bn": bool(bn),
                 "bias": bool(bias), "activation": "relu" if relu else None}
        self.trace.append(entry)
        if padding == "SAME":
            pt, pb = _same(x.shape[2], kh, stride)
            pl, pr = _same(x.shape[3], kw, stride)
            x = F.pad(x, (pl, pr, pt, pb))
        y = F.conv2d(x, w.permute(0, 3, 1, 2), stride=stride)
        if self.keep_acts and y.requires_grad:
            # debugging: the convolution's own input and output (+ its gradient dy), for the noise bounds of cancelling sums
            y.retain_grad()
            self.conv_io[scope] = (x.detach(), y)
        if bias:
            y = y + self.P[scope + "/biases"].view(1, -1, 1, 1)
        if bn:
            y = self.q(y)
            beta = self.P[scope + "/BatchNorm/beta"].view(1, -1, 1, 1)
            training = self.heads_bn_training if self._in_heads else self.bn_training
            if training:
                mean = y.mean((0, 2, 3))
                var = y.var((0, 2, 3), unbiased=False)
                mm, mv = self.P[scope + "/BatchNorm/moving_mean"], self.P[scope + "/BatchNorm/moving_variance"]
                self.new_moving[scope] = ((mm - (1 - self.bn_decay) * (mm - mean)).detach(),
                                          (mv - (1 - self.bn_decay) * (mv - var)).detach())
            else:
                mean, var = self.P[scope + "/BatchNorm/moving_mean"], self.P[scope + "/BatchNorm/moving_variance"]
            y = (y - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + BN_EPS) + beta
        if relu:
            y = torch.relu(y)
        if bn:
            y = self.q(y)
            if self.force is not None and scope in self.force:
                y = y + (self.force[scope] - y).detach()
            if self.keep_acts and y.requires_grad:
                y.retain_grad()
                self.acts[scope] = y
        entry["out_hw"] = [int(y.shape[2]), int(y.shape[3])]
        return _tag(y, [scope])

    def cat(self, ts):
        """tf.concat(3, ...) of model.py (channel axis)."""
        return _tag(torch.cat(ts, 1), sum((_prov(t) for t in ts), []))

    def max_pool(self, x, scope, k=3, stride=2):
        """slim.max_pool2d(k, stride, VALID)."""
        y = F.max_pool2d(x, k, stride)
        self.trace.append({"scope": scope, "op": "max_pool2d", "in_channels": int(x.shape[1]), "out_channels": int(y.shape[1]),
                           "kernel": [k, k], "stride": stride, "padding": "VALID", "inputs": _prov(x),
                           "out_hw": [int(y.shape[2]), int(y.shape[3])]})
        return _tag(y, [scope])

    def avg_pool(self, x, scope, k, padding):
        """slim.avg_pool2d(k, stride 1): SAME divides by the number of valid taps (model.py:134), VALID model.py:285."""
        y = F.avg_pool2d(x, k, 1, k // 2, count_include_pad=False) if padding == "SAME" else F.avg_pool2d(x, k, 1)
        y = self.q(y)
        self.trace.append({"scope": scope, "op": "avg_pool2d", "in_channels": int(x.shape[1]), "out_channels": int(y.shape[1]),
                           "kernel": [k, k], "stride": 1, "padding": padding, "inputs": _prov(x),
                           "out_hw": [int(y.shape[2]), int(y.shape[3])]})
        return _tag(y, [scope])

    def block(self, net, s, scale, relu, kind):
        """model.py:6-63."""
        c = self.conv
        if kind == 35:
            b0 = c(net, s + "Branch_0/Conv2d_1x1", 1)
            b1 = c(c(net, s + "Branch_1/Conv2d_0a_1x1", 1), s + "Branch_1/Conv2d_0b_3x3", 3)
            b2 = c(c(c(net, s + "Branch_2/Conv2d_0a_1x1", 1), s + "Branch_2/Conv2d_0b_3x3", 3), s + "Branch_2/Conv2d_0c_3x3", 3)
            mixed = self.cat([b0, b1, b2])
        elif kind == 17:
            b0 = c(net, s + "Branch_0/Conv2d_1x1", 1)
            b1 = c(c(c(net, s + "Branch_1/Conv2d_0a_1x1", 1), s + "Branch_1/Conv2d_0b_1x7", (1, 7)), s + "Branch_1/Conv2d_0c_7x1", (7, 1))
            mixed = self.cat([b0, b1])
        else:
            b0 = c(net, s + "Branch_0/Conv2d_1x1", 1)
            b1 = c(c(c(net, s + "Branch_1/Conv2d_0a_1x1", 1), s + "Branch_1/Conv2d_0b_1x3", (1, 3)), s + "Branch_1/Conv2d_0c_3x1", (3, 1))
            mixed = self.cat([b0, b1])
        up = c(mixed, s + "Conv2d_1x1", 1, bn=False, relu=False, bias=True)
        self.trace[-1]["residual"] = {"scale": scale, "skip": _prov(net), "activation": "relu" if relu else None}
        net = net + scale * up
        if relu:
            net = torch.relu(net)
        net = self.q(net)
        if self.force is not None and (s + "Conv2d_1x1") in self.force:
            net = net + (self.force[s + "Conv2d_1x1"] - net).detach()
        return _tag(net, [s + "Conv2d_1x1"])

    # The backbone as a chain of SEGMENTS (stem, Mixed_5b, each block35, Mixed_6a, each block17, Mixed_7a, each block8,
    # Block8, Conv2d_7b_1x1): backbone() composes them; tests/test_gpu_model.py runs them one at a time, back to front, for
    # the teacher-forced gradient check at BATCH_SIZE 64 (every segment boundary is a teacher-forcing point, so the chain
    # rule may be cut there: one segment's autograd graph in memory instead of the whole network's).
    def stem(self, x):
        """model.py:92-117: images [B,3,S,S] -> MaxPool_5a_3x3."""
        c, P = self.conv, "InceptionResnetV2/"
        net = c(x, P + "Conv2d_1a_3x3", 3, 2, "VALID")
        net = c(net, P + "Conv2d_2a_3x3", 3, 1, "VALID")
        net = c(net, P + "Conv2d_2b_3x3", 3)
        net = self.max_pool(net, P + "MaxPool_3a_3x3")
        net = c(net, P + "Conv2d_3b_1x1", 1, 1, "VALID")
        net = c(net, P + "Conv2d_4a_3x3", 3, 1, "VALID")
        return self.max_pool(net, P + "MaxPool_5a_3x3")

    def mixed_5b(self, net):
        """model.py:120-141."""
        c, Q = self.conv, "InceptionResnetV2/Mixed_5b/"
        b0 = c(net, Q + "Branch_0/Conv2d_1x1", 1)
        b1 = c(c(net, Q + "Branch_1/Conv2d_0a_1x1", 1), Q + "Branch_1/Conv2d_0b_5x5", 5)
        b2 = c(c(c(net, Q + "Branch_2/Conv2d_0a_1x1", 1), Q + "Branch_2/Conv2d_0b_3x3", 3), Q + "Branch_2/Conv2d_0c_3x3", 3)
        b3 = c(self.avg_pool(net, Q + "Branch_3/AvgPool_0a_3x3", 3, "SAME"), Q + "Branch_3/Conv2d_0b_1x1", 1)
        return self.cat([b0, b1, b2, b3])

    def mixed_6a(self, net):
        """model.py:145-161."""
        c, Q = self.conv, "InceptionResnetV2/Mixed_6a/"
        b0 = c(net, Q + "Branch_0/Conv2d_1a_3x3", 3, 2, "VALID")
        b1 = c(c(c(net, Q + "Branch_1/Conv2d_0a_1x1", 1), Q + "Branch_1/Conv2d_0b_3x3", 3), Q + "Branch_1/Conv2d_1a_3x3", 3, 2, "VALID")
        return self.cat([b0, b1, self.max_pool(net, Q + "Branch_2/MaxPool_1a_3x3")])

    def mixed_7a(self, net):
        """model.py:164-185."""
        c, Q = self.conv, "InceptionResnetV2/Mixed_7a/"
        b0 = c(c(net, Q + "Branch_0/Conv2d_0a_1x1", 1), Q + "Branch_0/Conv2d_1a_3x3", 3, 2, "VALID")
        b1 = c(c(net, Q + "Branch_1/Conv2d_0a_1x1", 1), Q + "Branch_1/Conv2d_1a_3x3", 3, 2, "VALID")
        b2 = c(c(c(net, Q + "Branch_2/Conv2d_0a_1x1", 1), Q + "Branch_2/Conv2d_0b_3x3", 3), Q + "Branch_2/Conv2d_1a_3x3", 3, 2, "VALID")
        return self.cat([b0, b1, b2, self.max_pool(net, Q + "Branch_3/MaxPool_1a_3x3")])

    def segments(self):
        """[(endpoint name or None, callable)] whose composition is backbone() (model.py:67-196)."""
        P = "InceptionResnetV2/"
        segs = [("MaxPool_5a_3x3", self.stem), ("Mixed_5b", self.mixed_5b)]
        for i in range(1, self.repeats[0] + 1):
            segs.append(("block35_10" if i == self.repeats[0] else None,
                         lambda net, i=i: self.block(net, P + "Repeat/block35_%d/" % i, 0.17, True, 35)))
        segs.append(("Mixed_6a", self.mixed_6a))
        for i in range(1, self.repeats[1] + 1):
            segs.append(("block17_20" if i == self.repeats[1] else None,
                         lambda net, i=i: self.block(net, P + "Repeat_1/block17_%d/" % i, 0.10, True, 17)))
        segs.append(("Mixed_7a", self.mixed_7a))
        for i in range(1, self.repeats[2] + 1):
            segs.append((None, lambda net, i=i: self.block(net, P + "Repeat_2/block8_%d/" % i, 0.20, True, 8)))
        segs.append((None, lambda net: self.block(net, P + "Block8/", 1.0, False, 8)))          # model.py:188
        segs.append(("Conv2d_7b_1x1", lambda net: self.conv(net, P + "Conv2d_7b_1x1", 1)))
        return segs

    def stage35(self, net):
        """model.py:120-142: Mixed_5b + block35 x repeats[0]."""
        for _, f in self.segments()[1:2 + self.repeats[0]]:
            net = f(net)
        return net

    def stage17(self, net):
        """model.py:145-162: Mixed_6a + block17 x repeats[1]."""
        a = 2 + self.repeats[0]
        for _, f in self.segments()[a:a + 1 + self.repeats[1]]:
            net = f(net)
        return net

    def stage8(self, net):
        """model.py:164-192: Mixed_7a + block8 x repeats[2] + Block8 (no relu, scale 1) + Conv2d_7b_1x1."""
        a = 3 + self.repeats[0] + self.repeats[1]
        for _, f in self.segments()[a:]:
            net = f(net)
        return net

    def backbone(self, x):
        """model.py:67-196.  x [B,3,S,S]."""
        net = x
        for name, f in self.segments():
            net = f(net)
            if name is not None:
                self.endpoints[name] = net
        return net

    def heads(self, feat):
        """model.py:198-324.  Returns raw locations [B,P,4] and confidence LOGITS [B,P]."""
        self._in_heads = True
        c, H, k = self.conv, "Multibox/", self.k
        out = lambda x, s, n: c(x, s, 1, bn=False, relu=False)
        nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], -1)
        locs, confs = [], []
        b = c(c(feat, H + "8x8/Conv", 1), H + "8x8/Conv_1", 3)
        locs.append(nhwc(out(b, H + "8x8/Conv_2", 4 * k))); confs.append(nhwc(out(b, H + "8x8/Conv_3", k)))
        b = c(c(feat, H + "6x6/Conv", 3), H + "6x6/Conv_1", 3, 1, "VALID")
        locs.append(nhwc(out(b, H + "6x6/Conv_2", 4 * k))); confs.append(nhwc(out(b, H + "6x6/Conv_3", k)))
        net = c(feat, H + "Conv", 3, 2)
        b = c(net, H + "4x4/Conv", 3)
        locs.append(nhwc(out(b, H + "4x4/Conv_1", 4 * k))); confs.append(nhwc(out(b, H + "4x4/Conv_2", k)))
        b = c(c(net, H + "3x3/Conv", 1), H + "3x3/Conv_1", 2, 1, "VALID")
        locs.append(nhwc(out(b, H + "3x3/Conv_2", 4 * k))); confs.append(nhwc(out(b, H + "3x3/Conv_3", k)))
        b = c(c(net, H + "2x2/Conv", 1), H + "2x2/Conv_1", 3, 1, "VALID")
        locs.append(nhwc(out(b, H + "2x2/Conv_2", 4 * k))); confs.append(nhwc(out(b, H + "2x2/Conv_3", k)))
        b = self.avg_pool(feat, H + "1x1/AvgPool2D", 8, "VALID")
        locs.append(nhwc(out(b, H + "1x1/Conv", 4))); confs.append(nhwc(out(b, H + "1x1/Conv_1", 1)))
        self._in_heads = False
        B = feat.shape[0]
        return torch.cat(locs, 1).reshape(B, -1, 4), torch.cat(confs, 1)

    def build(self, images_nhwc):
        """model.py:326-337: images [B,S,S,3] in [-1,1] -> (locations, logits); confidences = sigmoid(logits)."""
        x = _tag(self.q(images_nhwc).permute(0, 3, 1, 2), ["inputs"])
        return self.heads(self.backbone(x))


def multibox_loss(locs, logits, priors, gt, match, alpha):
    """loss.py:67-101 in torch given the (constant) matching; returns (loc_loss, conf_loss)."""
    B, P = logits.shape
    dec = locs + priors.unsqueeze(0)
    c = torch.sigmoid(logits) + 1e-10
    m = torch.as_tensor(match, dtype=torch.long)
    pos = m >= 0
    gsel = torch.gather(gt, 1, m.clamp(min=0).unsqueeze(-1).expand(B, P, 4))
    loc = alpha * 0.5 * (((dec - gsel) ** 2) * pos.unsqueeze(-1)).sum()
    conf = -(torch.log(c) * pos).sum() - (torch.log((1.0 - c) + 1e-10) * (~pos)).sum()
    return loc, conf
