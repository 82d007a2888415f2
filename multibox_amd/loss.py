"""Matching + loss -- host mirror of the reference's loss.py on libmbx.

``compute_assignments`` and ``add_loss`` keep the reference's names and argument meaning
(loss.py:8, loss.py:55); tensors are torch CUDA tensors instead of TF tensors.
``MultiboxLoss`` is the fused training entry (decode -> match -> loss + gradients), all
on the caller's stream, graph-capturable.
"""
from __future__ import annotations

import torch

from . import _lib

SMALL_EPSILON = 1e-10   # loss.py:6


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32(t):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expected contiguous float32 CUDA tensor"
    return t


def _i32(t):
    assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous(), "expected contiguous int32 CUDA tensor"
    return t


def match_boxes(decoded, conf, gt_bboxes, num_gt_bboxes, alpha, match=None, status=None):
    """mbx_match: decoded [B,P,4], conf [B,P] (+1e-10 applied), gt [B,G,4], n [B] -> match int32 [B,P]."""
    B, P = decoded.shape[0], decoded.shape[1]
    G = gt_bboxes.shape[1]
    if match is None:
        match = torch.empty((B, P), dtype=torch.int32, device=decoded.device)
    if status is None:
        status = torch.empty((B,), dtype=torch.int32, device=decoded.device)
    _lib.check(_lib.lib().mbx_match(_f32(decoded).data_ptr(), _f32(conf).data_ptr(), _f32(gt_bboxes).data_ptr(),
                                    _i32(num_gt_bboxes).data_ptr(), float(alpha), B, P, G, match.data_ptr(),
                                    status.data_ptr(), None, 0, _stream()), "mbx_match")
    return match, status


def compute_assignments(locations, confidences, gt_bboxes, num_gt_bboxes, batch_size, alpha):
    """loss.py:8-53.  Same inputs/outputs as the py_func callback: returns
    [assignment_partitions int32 [B*P], stacked_gt_bboxes float32 [M,4]] (row order),
    and raises like the callback would (scipy ValueError -> TF UnknownError) on a bad cost matrix."""
    B = int(batch_size)
    loc = locations.reshape(B, -1, 4).contiguous()
    P = loc.shape[1]
    conf = confidences.reshape(B, P).contiguous()
    match, status = match_boxes(loc, conf, gt_bboxes.contiguous(), num_gt_bboxes.to(torch.int32).contiguous(), alpha)
    st = status.cpu()
    if int(st.max()) != 0:
        raise ValueError("compute_assignments: infeasible or non-finite cost matrix (status %s)" % st.tolist())
    part = (match >= 0).to(torch.int32).reshape(-1)
    b_idx, p_idx = torch.nonzero(match >= 0, as_tuple=True)           # row-major == ascending row order
    stacked = gt_bboxes[b_idx, match[b_idx, p_idx].long()].to(torch.float32)
    return [part, stacked]


class MultiboxLoss:
    """Fused decode + match + loss (+ gradients) with preallocated buffers (no allocation per step)."""

    def __init__(self, bbox_priors, batch_size, max_num_bboxes, location_loss_alpha, device="cuda"):
        self.priors = torch.as_tensor(bbox_priors, dtype=torch.float32).to(device).contiguous()
        self.P = self.priors.shape[0]
        self.B, self.G, self.alpha = int(batch_size), int(max_num_bboxes), float(location_loss_alpha)
        f = dict(dtype=torch.float32, device=device)
        self.decoded = torch.empty((self.B, self.P, 4), **f)
        self.conf = torch.empty((self.B, self.P), **f)
        self.match = torch.empty((self.B, self.P), dtype=torch.int32, device=device)
        self.status = torch.zeros((self.B,), dtype=torch.int32, device=device)
        self.loss2 = torch.zeros((2,), **f)
        self.d_locs = torch.empty((self.B, self.P, 4), **f)
        self.d_logits = torch.empty((self.B, self.P), **f)
        self.ws = torch.empty((_lib.lib().mbx_loss_workspace_bytes(self.B),), dtype=torch.uint8, device=device)

    def forward_backward(self, raw_locs, logits, gt_bboxes, num_gt_bboxes, grad_scale=1.0, conf_is_logit=True):
        """raw_locs [B,P,4], logits [B,P] f32 -> (loss2 [2] = {location_loss, confidence_loss}, d_locs, d_logits)."""
        l, s = _lib.lib(), _stream()
        B, P, G = self.B, self.P, self.G
        assert raw_locs.shape == (B, P, 4) and logits.numel() == B * P and gt_bboxes.shape == (B, G, 4)
        if conf_is_logit:
            _lib.check(l.mbx_decode_conf(_f32(raw_locs).data_ptr(), _f32(logits).data_ptr(), self.priors.data_ptr(),
                                         B, P, SMALL_EPSILON, self.decoded.data_ptr(), self.conf.data_ptr(), s),
                       "mbx_decode_conf")
        else:
            _lib.check(l.mbx_decode_conf(_f32(raw_locs).data_ptr(), None, self.priors.data_ptr(), B, P, 0.0,
                                         self.decoded.data_ptr(), None, s), "mbx_decode_conf")
            torch.add(logits.reshape(B, P), SMALL_EPSILON, out=self.conf)        # loss.py:74
        _lib.check(l.mbx_match(self.decoded.data_ptr(), self.conf.data_ptr(), _f32(gt_bboxes).data_ptr(),
                               _i32(num_gt_bboxes).data_ptr(), self.alpha, B, P, G, self.match.data_ptr(),
                               self.status.data_ptr(), None, 0, s), "mbx_match")
        _lib.check(l.mbx_loss_fwd_bwd(self.decoded.data_ptr(), _f32(logits).data_ptr(), int(bool(conf_is_logit)),
                                      gt_bboxes.data_ptr(), self.match.data_ptr(), self.alpha, float(grad_scale),
                                      B, P, G, self.loss2.data_ptr(), self.d_locs.data_ptr(), self.d_logits.data_ptr(),
                                      self.ws.data_ptr(), self.ws.numel(), s), "mbx_loss_fwd_bwd")
        return self.loss2, self.d_locs, self.d_logits


def add_loss(locations, confidences, batched_bboxes, batched_num_bboxes, bbox_priors, location_loss_alpha):
    """loss.py:55-116 (forward value): locations [B,P,4] raw residuals, confidences [B,P,1] sigmoid outputs.
    Returns (location_loss, confidence_loss) as 0-d CUDA tensors."""
    B = locations.shape[0]
    m = MultiboxLoss(bbox_priors, B, batched_bboxes.shape[1], location_loss_alpha, device=locations.device)
    loss2, _, _ = m.forward_backward(locations.contiguous(), confidences.reshape(B, -1).contiguous(),
                                     batched_bboxes.to(torch.float32).contiguous(),
                                     batched_num_bboxes.to(torch.int32).contiguous(), conf_is_logit=False)
    st = m.status.cpu()
    if int(st.max()) != 0:
        raise ValueError("add_loss: matching failed (status %s)" % st.tolist())
    return loss2[0].clone(), loss2[1].clone()
