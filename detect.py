#!/usr/bin/env python3
"""Drop-in for the reference's detect.py (detect.py:462-524): same flags and results JSON
(`results-dense-<global_step>.json`, detect.py:438-460).  Forward pass + decode/clip/filter/
top-K/convert all run on the GPU (libmbx); NO NMS, exactly like the reference.

`--tfrecords` runs the reference's multi-crop input (original / flipped / sliding crops per
config.yaml DETECTION, detect.py:134-292) through multibox_amd/inputs.py (host, PIL JPEG decode, TF-0.11
bilinear resize); `--synthetic N` feeds N seeded synthetic patches with whole-image metadata instead."""
import argparse
import json
import os
import pprint
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args():
    p = argparse.ArgumentParser(description="Detect objects using a pretrained Multibox model")
    p.add_argument("--tfrecords", dest="tfrecords", type=str, nargs="+", required=False, default=None)
    p.add_argument("--priors", dest="priors", type=str, required=True)
    p.add_argument("--checkpoint_path", dest="checkpoint_path", type=str, required=True)
    p.add_argument("--config", dest="config_file", type=str, required=True)
    p.add_argument("--max_iterations", dest="max_iterations", type=int, default=0)
    p.add_argument("--max_detections", dest="max_detections", type=int, default=100,
                   help="accepted for compatibility; unused by the reference too (detect.py:294)")
    p.add_argument("--save_dir", dest="save_dir", type=str, required=True)
    p.add_argument("--synthetic", type=int, default=0, help="[new] number of synthetic images instead of --tfrecords")
    p.add_argument("--keep_partial_batch", action="store_true",
                   help="[new] also process the last incomplete batch (the reference's tf.train.batch drops it)")
    return p.parse_args()


def main():
    args = parse_args()
    import numpy as np
    import torch
    from multibox_amd.config import parse_config_file, with_defaults
    from multibox_amd import priors as PR, checkpoint as CK, detect as D, _lib
    from multibox_amd.engine import Net
    import __graft_entry__ as g
    # one process per GPU under torch.distributed.run: ranks take disjoint batches, rank 0 writes the one JSON
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        print("Command line arguments:")
        pprint.pprint(vars(args))
    cfg = with_defaults(parse_config_file(args.config_file))
    if not args.tfrecords and not args.synthetic:
        raise SystemExit("give --tfrecords FILE... or --synthetic N")
    torch.cuda.set_device(local_rank)
    if world > 1:       # results are gathered as host objects: gloo is enough, no GPU collective on this path
        torch.distributed.init_process_group("gloo")
    if local_rank == 0:
        g.build()
    if world > 1:
        torch.distributed.barrier()
    bbox_priors = PR.load_priors(args.priors)
    ckpt = CK.latest_checkpoint(args.checkpoint_path)
    if ckpt is None:
        print("ERROR: No checkpoint file found.")
        return
    B = cfg.BATCH_SIZE
    net = Net(batch=B, input_size=cfg.INPUT_SIZE, k=cfg.NUM_BBOXES_PER_CELL, mode="infer")
    global_step = CK.restore_for_inference(ckpt, net)
    print("Found model for global step: %d" % global_step)
    det = cfg.get("DETECTION", {})
    keeps = [int(det.get("ORIGINAL_IMAGE_MAX_TO_KEEP", 200)), int(det.get("FLIPPED_IMAGE_MAX_TO_KEEP", 0))]
    keeps += [int(c.MAX_TO_KEEP) for c in (det.get("CROPS", None) or [])]
    max_keep = max(keeps + [1])
    # DETECTION.NMS_IOU_THRESHOLD (not a key of the reference, which has no NMS -- detect.py:408-443): optional greedy
    # per-patch non-maximum suppression after the top-K stage; absent = the reference's behaviour
    nms_iou = det.get("NMS_IOU_THRESHOLD", None)
    pp = D.DetectPostprocess(bbox_priors, B, k_max=max_keep, nms_iou=nms_iou)
    conf = torch.empty((B, net.P), dtype=torch.float32, device="cuda")
    results, step = [], 0
    S = cfg.INPUT_SIZE

    def synthetic_batches():
        n_images = args.synthetic
        for start in range(0, n_images - n_images % B if n_images >= B else 0, B):
            rng = np.random.RandomState(start)
            yield dict(images=rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32), offsets=np.zeros((B, 2), np.int32),
                       dims=np.tile([[S, S]], (B, 1)), is_flipped=np.zeros((B, 1), np.int32),
                       restrictions=np.tile([[0., 0., 1., 1.]], (B, 1)), max_to_keep=np.full((B, 1), keeps[0]),
                       image_hw=np.tile([[S, S]], (B, 1)), image_ids=list(range(start, start + B)))
    if args.tfrecords and not args.synthetic:
        from multibox_amd.inputs import detect_batches
        from multibox_amd.augment import PatchExtractor
        # the host decodes and lays out the patches (detect.py:183-281); scaling + bilinear resize run on the GPU
        on_device = bool(cfg.get("INPUT_AUGMENT_ON_DEVICE", True))
        depth = int(cfg.get("INPUT_PREFETCH_BATCHES", 3))
        extractor = PatchExtractor(B, S, slots=depth + 2) if on_device else None
        # rank-sharded INPUT: this rank decodes only the records that feed its own batches
        in_stats = {}
        # produced by a background thread a few batches ahead (record parsing, patch planning and batch assembly are host
        # work that would otherwise sit between two forward passes): INPUT_PREFETCH_BATCHES, 0 = in this thread
        from multibox_amd.inputs import prefetched
        stream = detect_batches(args.tfrecords, cfg, B, keep_partial=args.keep_partial_batch, device_patches=on_device,
                                rank=rank, world=world, stats=in_stats)
        if on_device:
            # the HOST half of the patch extraction (24 MB of pixels per batch into pinned staging, the item table) runs in the
            # producer thread too; this thread only uploads and launches (PatchExtractor.prepare / launch)
            def staged(it):
                for b in it:
                    b["staged"] = extractor.prepare(b.pop("sources"), b.pop("patches"))
                    yield b
            stream = staged(stream)
        batches = ((b["batch_index"], b) for b in (prefetched(stream, depth) if depth > 0 else stream))
    else:
        in_stats, extractor = None, None
        batches = D.shard_batches(synthetic_batches(), rank, world)
    # Software pipeline: batch i+1 (input upload, patch extraction, forward, decode / filter / top-K, D2H of its results
    # into pinned buffers) is ENQUEUED before the host turns batch i's results into records, so the Python loop of
    # detect.py:408-443 runs while the GPU works.  Events, not synchronize(), bound each batch.
    host_out = [(torch.empty((B, max_keep, 4), dtype=torch.float64, pin_memory=True),
                 torch.empty((B, max_keep), dtype=torch.float32, pin_memory=True),
                 torch.empty((B,), dtype=torch.int32, pin_memory=True)) for _ in range(2)]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(2)]      # start, device work done, copied
    pending = None

    # The device work of one batch -- pack the input, 205 forward launches, sigmoid, decode / clip / filter / top-K -- is
    # captured ONCE into a hipGraph over static buffers (its ~210 eager launches cost 2.6 ms of host time per batch of 64,
    # more than half of the 4.7 ms the GPU needs): per batch the host fills the input picture buffer and the patch
    # metadata, replays the graph and queues the copies of the results.  DETECT_HIP_GRAPH: false = launch eagerly.
    import ctypes
    x_static = extractor.out if extractor is not None else torch.empty((B, S, S, 3), dtype=torch.float32, device="cuda")
    meta_bytes = B * ctypes.sizeof(_lib.PatchMeta)
    meta_static = torch.zeros(meta_bytes, dtype=torch.uint8, device="cuda")
    meta_host = [torch.empty(meta_bytes, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    out_static = {}

    def device_step():
        net.set_input(x_static)
        locs, logits = net.forward()
        _lib.check(_lib.lib().mbx_decode_conf(None, logits.data_ptr(), None, B, net.P, 0.0, None, conf.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "sigmoid")
        out_static["r"] = pp(locs, conf, meta_static)
    graph = None
    if bool(cfg.get("DETECT_HIP_GRAPH", True)):
        x_static.zero_()
        device_step()                                   # warm-up (lazy module loads) on an all-zero batch
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            device_step()

    # The record text (float repr of ~17 000 numbers per batch: ~10 ms of pure host work) is produced by worker
    # PROCESSES (NUM_RECORD_WORKERS, default 3; spawned: no fork after the GPU is initialised; numpy + json only), so this
    # thread only feeds the GPU.  0 = in this process.
    n_workers = int(cfg.get("NUM_RECORD_WORKERS", 3))
    pool = None
    if n_workers > 0:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        pool = ProcessPoolExecutor(max_workers=n_workers, mp_context=mp.get_context("spawn"))
    from multibox_amd import records as REC

    def finish(p):
        slot, bi_, ids_ = p
        ev[slot][2].synchronize()
        hb, hs, hc = host_out[slot]
        if pool is not None:
            results.append((bi_, pool.submit(REC.batch_chunk, hb.numpy().copy(), hs.numpy().copy(), hc.numpy().copy(), ids_)))
        else:
            results.append((bi_, REC.batch_chunk(hb.numpy(), hs.numpy(), hc.numpy(), ids_)))
        print("Step: %d, Time/image (ms): %.2f" % (len(results), ev[slot][0].elapsed_time(ev[slot][1]) / B))   # detect.py:446

    t_all = time.time()
    for bi, batch in batches:
        slot = step % 2
        meta = D.make_patch_meta(batch["offsets"], batch["dims"], batch["is_flipped"], batch["restrictions"],
                                 batch["max_to_keep"], batch["image_hw"], device="cpu")
        meta_host[slot].copy_(meta)                     # slot's previous upload finished before its results were read (finish)
        ev[slot][0].record()
        if "staged" in batch:
            extractor.launch(batch["staged"])                                   # -> x_static (its own output buffer)
        else:
            x_static.copy_(torch.from_numpy(batch["images"]), non_blocking=False)
        meta_static.copy_(meta_host[slot], non_blocking=True)
        if graph is not None:
            graph.replay()
        else:
            device_step()
        ev[slot][1].record()
        boxes, scores, _, count = out_static["r"]
        for h, d in zip(host_out[slot], (boxes, scores, count)):
            h.copy_(d, non_blocking=True)
        ev[slot][2].record()
        ids = [int(i) if str(i).lstrip("-").isdigit() else i for i in batch["image_ids"]]     # detect.py:410 int(image_id)
        if pending is not None:
            finish(pending)
        pending = (slot, bi, ids)
        step += 1
        if args.max_iterations > 0 and step == args.max_iterations:
            break
    if pending is not None:
        finish(pending)
    if step:
        print("rank %d: %d patches in %.2f s (%.0f patches/s, input + forward + post-process + records)%s" % (
            rank, step * B, time.time() - t_all, step * B / max(time.time() - t_all, 1e-9),
            "" if not in_stats else "; decoded %d of %d records" % (in_stats.get("decoded", 0), in_stats.get("records", 0))))
    n_records = 0
    for i, (bi_, r) in enumerate(results):           # collect the workers' chunks: (batch index, [records joined with ", "])
        n, text = r.result() if pool is not None else r
        n_records += n
        results[i] = (bi_, [text])
    if pool is not None:
        pool.shutdown()
    results = D.gather_results(results)
    if world > 1:
        cnt = torch.tensor([n_records], dtype=torch.int64)
        torch.distributed.all_reduce(cnt)
        n_records = int(cnt)
    if rank == 0:
        os.makedirs(args.save_dir, exist_ok=True)
        save_path = os.path.join(args.save_dir, "results-dense-%d.json" % global_step)
        with open(save_path, "w") as f:
            f.write(D.records_to_json(results))        # the same text as json.dump(list of record dicts, f)
        print("wrote", save_path, n_records, "detections")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
