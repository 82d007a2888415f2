#!/usr/bin/env python3
"""Drop-in for the reference's eval.py (eval.py:248-305): same flags.  Forward pass in inference mode with the EMA
weights (eval.py:46-76), decode + clip + top-100 on the GPU (eval.py:144-173 -> mbx_decode_filter_topk with
whole-image metadata), boxes scaled to INPUT_SIZE pixels, then COCO bbox AP/AR with useCats = 0 (eval.py:212-226;
multibox_amd/cocoeval.py restates pycocotools' COCOeval, which is not installed here).  The twelve summary numbers are
printed in COCOeval's format and written to <summary_dir>/eval-<global_step>.json (the reference writes a TF event
file, eval.py:228-246)."""
import argparse
import json
import os
import pprint
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args():
    p = argparse.ArgumentParser(description="Evaluate a Multibox model with the COCO bounding-box metric")
    p.add_argument("--tfrecords", dest="tfrecords", type=str, nargs="+", required=True)
    p.add_argument("--priors", dest="priors", type=str, required=True)
    p.add_argument("--summary_dir", dest="summary_dir", type=str, required=True)
    p.add_argument("--checkpoint_path", dest="checkpoint_path", type=str, required=True)
    p.add_argument("--config", dest="config_file", type=str, required=True)
    p.add_argument("--max_iterations", dest="max_iterations", type=int, default=0)
    return p.parse_args()


def main():
    args = parse_args()
    import numpy as np
    import torch
    from multibox_amd.config import parse_config_file, with_defaults
    from multibox_amd import priors as PR, checkpoint as CK, detect as D, _lib
    from multibox_amd.cocoeval import evaluate_bbox
    from multibox_amd.engine import Net
    from multibox_amd.inputs import eval_batches
    import __graft_entry__ as g
    print("Command line arguments:")
    pprint.pprint(vars(args))
    cfg = with_defaults(parse_config_file(args.config_file))
    g.build()
    bbox_priors = PR.load_priors(args.priors)
    ckpt = CK.latest_checkpoint(args.checkpoint_path)
    if ckpt is None:
        print("ERROR: No checkpoint file found.")
        return
    B, S = cfg.BATCH_SIZE, cfg.INPUT_SIZE
    net = Net(batch=B, input_size=S, k=cfg.NUM_BBOXES_PER_CELL, mode="infer")
    global_step = CK.restore_for_inference(ckpt, net)
    print("Found model for global step: %d" % global_step)
    K = 100                                                                      # eval.py:166 `for k in range(100)`
    pp = D.DetectPostprocess(bbox_priors, B, k_max=K)
    # whole image, no restriction, not flipped, image = patch = INPUT_SIZE: decoded boxes stay normalised
    meta = D.make_patch_meta(np.zeros((B, 2), np.int32), np.tile([[S, S]], (B, 1)), np.zeros((B, 1), np.int32),
                             np.tile([[0., 0., 1., 1.]], (B, 1)), np.full((B, 1), K), np.tile([[S, S]], (B, 1)))
    conf = torch.empty((B, net.P), dtype=torch.float32, device="cuda")
    gt_annotations, pred_annotations, gt_id, step = [], [], 1, 0
    on_device = bool(cfg.get("INPUT_AUGMENT_ON_DEVICE", True))      # legacy bilinear resize + [-1,1] scaling on the GPU
    if on_device:
        from multibox_amd.augment import BatchAugmenter
        aug = BatchAugmenter(B, S, slot_bytes=1024 * 1024 * 3)
    # the device work of a batch (pack, forward, sigmoid, decode / top-100) as ONE hipGraph over static buffers, as in detect.py
    x_static = torch.zeros((B, S, S, 3), dtype=torch.float32, device="cuda")
    out_static = {}

    def device_step():
        net.set_input(x_static)
        locs, logits = net.forward()
        _lib.check(_lib.lib().mbx_decode_conf(None, logits.data_ptr(), None, B, net.P, 0.0, None, conf.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "sigmoid")
        out_static["r"] = pp(locs, conf, meta)
    device_step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        device_step()
    for images, gt, n_gt, areas, ids in eval_batches(args.tfrecords, cfg, B, cfg.MAX_NUM_BBOXES, device_images=on_device):
        t = time.time()
        if on_device:
            aug.begin()
            for u8 in images:
                aug.add(u8, 0, False, [])
            x_static.copy_(aug.run())
        else:
            x_static.copy_(torch.from_numpy(images))
        graph.replay()
        boxes, scores, _, count = out_static["r"]
        torch.cuda.synchronize()
        dt = time.time() - t
        boxes, scores, count = boxes.cpu().numpy() * S, scores.cpu().numpy(), count.cpu().numpy()     # eval.py:158-160
        for b in range(B):
            img_id = int(ids[b])                                                                       # eval.py:142
            for k in range(int(count[b])):
                x1, y1, x2, y2 = boxes[b, k]
                pred_annotations.append([img_id, x1, y1, x2 - x1, y2 - y1, float(scores[b, k]), 1])
            for k in range(int(n_gt[b])):
                x1, y1, x2, y2 = (gt[b, k] * S).tolist()
                gt_annotations.append({"id": gt_id, "image_id": img_id, "category_id": 1, "area": float(areas[b, k]),
                                       "bbox": [x1, y1, x2 - x1, y2 - y1], "iscrowd": 0})
                gt_id += 1
        step += 1
        print("Step: %d, Time/image (ms): %.1f" % (step, dt / B * 1000))
        if args.max_iterations > 0 and step == args.max_iterations:
            break
    stats, lines = evaluate_bbox(gt_annotations, pred_annotations)
    out = {}
    for line in lines:
        print(line)
        description, score = line.rsplit("=", 1)
        out[description.strip()] = float(score)
    os.makedirs(args.summary_dir, exist_ok=True)
    path = os.path.join(args.summary_dir, "eval-%d.json" % global_step)
    with open(path, "w") as f:
        json.dump({"global_step": global_step, "images": step * B, "stats": stats, "summary": out}, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
