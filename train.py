#!/usr/bin/env python3
"""Drop-in for the reference's train.py (train.py:300-385): same flags, same config.yaml keys,
same priors.pkl; the step runs on libmbx (one process per GPU; launch with torch.distributed.run
for data parallelism).

`--tfrecords` reads the reference's TFRecords (inputs.py:225-247) through multibox_amd/inputs.py, augmentations
included (bbox shift, distorted crop, random resize method, colour distortion, flip: inputs.py:44-203) -- SURVEY 8f F1;
`--synthetic` feeds seeded synthetic batches of the same input contract (inputs.py:340-351).  Under
torch.distributed.run BATCH_SIZE is the per-GPU batch: gradients are summed over ranks (the reference loss is a batch
sum), so the learning-rate schedule counts BATCH_SIZE * world images per step.  --pretrained_model takes one of this build's .pt files or a TensorFlow V1 checkpoint (multibox_amd/tf_checkpoint.py, F2)."""
import argparse
import json
import os
import pprint
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args():
    p = argparse.ArgumentParser(description="Train the multibox detection system")
    p.add_argument("--tfrecords", dest="tfrecords", type=str, nargs="+", required=False, default=None,
                   help="paths to tfrecords files that contain the training data")
    p.add_argument("--priors", dest="priors", type=str, required=True, help="path to the bounding box priors pickle file")
    p.add_argument("--logdir", dest="logdir", type=str, required=True, help="path to directory to store summary files and checkpoint files")
    p.add_argument("--config", dest="config_file", type=str, required=True, help="Path to the configuration file")
    p.add_argument("--pretrained_model", dest="pretrained_model", type=str, default=None)
    p.add_argument("--fine_tune", dest="fine_tune", action="store_true", default=False,
                   help="only the variables in the detection heads will be trained")
    p.add_argument("--trainable_scopes", dest="trainable_scopes", type=str, nargs="+", default=None)
    p.add_argument("--use_moving_averages", dest="use_moving_averages", action="store_true", default=False)
    p.add_argument("--restore_moving_averages", dest="restore_moving_averages", action="store_true", default=False)
    p.add_argument("--max_number_of_steps", dest="max_number_of_steps", type=int, default=None)
    p.add_argument("--batch_size", dest="batch_size", type=int, default=None)
    p.add_argument("--synthetic", action="store_true", help="[new] synthetic input instead of --tfrecords")
    return p.parse_args()


def main():
    args = parse_args()
    import numpy as np
    import torch
    from multibox_amd.config import parse_config_file, with_defaults
    from multibox_amd import priors as PR, checkpoint as CK
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer, decay_steps
    from multibox_amd.synth import synthetic_batch
    from multibox_amd.dist import bn_max_workgroups_for
    import __graft_entry__ as g

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        print("Command line arguments:")
        pprint.pprint(vars(args))
    cfg = with_defaults(parse_config_file(args.config_file))
    if args.max_number_of_steps is not None:      # train.py:362-366
        cfg.NUM_TRAIN_ITERATIONS = args.max_number_of_steps
    if args.batch_size is not None:
        cfg.BATCH_SIZE = args.batch_size
    if not args.tfrecords and not args.synthetic:
        raise SystemExit("give --tfrecords FILE... or --synthetic")
    if args.pretrained_model and not os.path.exists(args.pretrained_model):
        raise SystemExit("pretrained model not found: %s" % args.pretrained_model)
    torch.cuda.set_device(local_rank)
    pg = None
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        pg = torch.distributed.group.WORLD
    if local_rank == 0:                       # one build per node (the .so lives in the node's copy of the tree)
        g.build()
    if world > 1:
        torch.distributed.barrier()
    bbox_priors = PR.load_priors(args.priors)                     # train.py:368-370
    net = Net(batch=cfg.BATCH_SIZE, input_size=cfg.INPUT_SIZE, k=cfg.NUM_BBOXES_PER_CELL, mode="train",
              fine_tune=args.fine_tune, bn_decay=cfg.BATCHNORM_MOVING_AVERAGE_DECAY,
              bn_max_workgroups=bn_max_workgroups_for(world, torch.cuda.get_device_properties(local_rank).multi_processor_count)[0])
    tr = Trainer(net, bbox_priors, max_num_bboxes=cfg.MAX_NUM_BBOXES, location_loss_alpha=cfg.LOCATION_LOSS_ALPHA,
                 initial_learning_rate=cfg.INITIAL_LEARNING_RATE,
                 decay_steps_=decay_steps(cfg.NUM_TRAIN_EXAMPLES, cfg.BATCH_SIZE * world, cfg.NUM_EPOCHS_PER_DELAY),
                 learning_rate_decay_factor=cfg.LEARNING_RATE_DECAY_FACTOR, staircase=cfg.LEARNING_RATE_STAIRCASE,
                 rmsprop_decay=cfg.RMSPROP_DECAY, rmsprop_momentum=float(cfg.RMSPROP_MOMENTUM), rmsprop_epsilon=cfg.RMSPROP_EPSILON,
                 moving_average_decay=cfg.MOVING_AVERAGE_DECAY, process_group=pg, trainable_scopes=args.trainable_scopes)
    if args.trainable_scopes and rank == 0:       # train.py:166-169
        print("Trainable Variables")
        for name in tr.trainable_names:
            print(name)
    latest = CK.latest_checkpoint(args.logdir)                    # slim.learning.train resumes from logdir
    if latest and latest.endswith(".pt"):
        CK.restore_for_training(latest, tr)
        if rank == 0:
            print("Resumed from %s (step %d)" % (latest, tr.global_step))
    elif args.pretrained_model:
        ck = CK.restore_pretrained(args.pretrained_model, tr, fine_tune=args.fine_tune,            # train.py:15-90
                                   use_moving_averages=args.use_moving_averages,
                                   restore_moving_averages=args.restore_moving_averages)
        if rank == 0:
            print("Initialised from %s" % ck)
    tr.broadcast_parameters(src=0)                # every rank starts from rank 0's variables, slots and shadows
    t_save = t_log = time.time()
    log = open(os.path.join(args.logdir, "train_log.jsonl"), "a") if rank == 0 and (os.makedirs(args.logdir, exist_ok=True) or True) else None
    step0 = tr.global_step
    real = None
    if args.tfrecords and not args.synthetic:
        from multibox_amd.input_workers import ParallelTrainInput, DevicePrefetcher
        files = args.tfrecords[rank::world] if len(args.tfrecords) >= world else args.tfrecords     # shard files over ranks
        # NUM_INPUT_THREADS worker processes (inputs.py:353-371) -> shared-memory ring -> pinned buffers -> async H2D
        src = ParallelTrainInput(files, cfg, cfg.BATCH_SIZE, cfg.MAX_NUM_BBOXES, num_workers=int(cfg.get("NUM_INPUT_THREADS", 4)),
                                 num_epochs=None, seed=int(cfg.get("RANDOM_SEED", 1)) + 1000 * rank, shuffle=True,   # train.py:214-225
                                 capacity=int(cfg.get("QUEUE_CAPACITY", 1000)), min_after_dequeue=int(cfg.get("QUEUE_MIN", 96)),
                                 device_augment=bool(cfg.get("INPUT_AUGMENT_ON_DEVICE", True)))   # resize / colour / flip on the GPU
        # The augmentation kernels are launched in front of each step on the TRAINING stream (0.26 ms per batch; only the
        # H2D upload rides the prefetcher's side stream).  INPUT_AUGMENT_KERNELS_ON_SIDE_STREAM: true is an explicit opt-in
        # to overlapping them with the step: they then compete for CUs with the one-workgroup-per-CU grid barrier of the
        # BN backward (a timeout falls back to the three-launch form, Trainer.check_health) -- +0.5 % throughput at best.
        real = DevicePrefetcher(src, cfg.BATCH_SIZE, cfg.INPUT_SIZE, cfg.MAX_NUM_BBOXES, device="cuda", depth=2,
                                kernels_on_main=not bool(cfg.get("INPUT_AUGMENT_KERNELS_ON_SIDE_STREAM", False)))

    exhausted = [False]

    def next_real():
        # An exhausted / empty file shard on ONE rank must stop every rank, not leave the others in all_reduce -- without
        # a host sync per step: the rank raises the stop word of the step control block (Trainer.request_stop), which
        # is summed over ranks with the gradients and makes every rank's optimiser skip from that step on; all ranks
        # learn of it at the next health check (LOG_EVERY_N_STEPS) and leave together.  Until then this rank re-runs
        # its last batch (the steps are not applied).
        if not exhausted[0]:
            try:
                return real.next()
            except StopIteration:
                exhausted[0] = True
                if world == 1:
                    real.close()
                    print("saved", CK.save(args.logdir, tr, cfg.MAX_TO_KEEP, cfg.get("KEEP_CHECKPOINT_EVERY_N_HOURS", 10000.0)))
                    raise SystemExit("input exhausted at step %d" % tr.global_step)
                tr.request_stop()
        return None

    while tr.global_step < cfg.NUM_TRAIN_ITERATIONS:
        if real is not None:
            b = next_real()                                     # already on the device (prefetched two batches ahead)
            if b is not None:
                tr.set_batch(b[0], b[1], b[2])
        else:
            images, gt, n = synthetic_batch(cfg.BATCH_SIZE, cfg.INPUT_SIZE, cfg.MAX_NUM_BBOXES, seed=tr.global_step * world + rank)
            tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
        tr.step()
        log_now = tr.global_step % cfg.LOG_EVERY_N_STEPS == 0  # (decided BEFORE the health check takes skipped steps off global_step)
        if log_now:                                           # (a collective: the same steps on every rank)
            # every rank: matching status (py_func error -> abort, loss.py:82), barrier timeouts (-> in-process fall-back
            # to the three-launch BN backward), weight-gradient work tallies, stop requests
            health = tr.check_health()
            if log is not None:
                while tr.events:
                    log.write(json.dumps(tr.events.pop(0)) + "\n")
            if health["stop"]:
                if real is not None:
                    real.close()
                if rank == 0:                                 # slim.learning.train saves when the input runs out (OutOfRange)
                    print("saved", CK.save(args.logdir, tr, cfg.MAX_TO_KEEP, cfg.get("KEEP_CHECKPOINT_EVERY_N_HOURS", 10000.0)))
                if world > 1:
                    torch.distributed.destroy_process_group()
                raise SystemExit("input exhausted on at least one rank; stopped at step %d (applied steps)" % tr.global_step)
        if rank == 0 and log_now:
            loc, conf, reg, total = tr.losses()
            now = time.time()
            ips = cfg.BATCH_SIZE * world * (tr.global_step - step0) / (now - t_log) if now > t_log else 0.0
            rec = dict(global_step=tr.global_step, total_loss=total, location_loss=loc, confidence_loss=conf,
                       learning_rate=tr.lr, images_per_sec=ips)          # train.py:266-271 summaries
            print("global step %d: loss = %.4f (loc %.4f conf %.4f) lr %.6f %.1f img/s" % (tr.global_step, total, loc, conf, tr.lr, ips))
            log.write(json.dumps(rec) + "\n")
            log.flush()
            step0, t_log = tr.global_step, now
        if rank == 0 and time.time() - t_save > cfg.SAVE_INTERVAL_SECS:
            CK.save(args.logdir, tr, cfg.MAX_TO_KEEP, cfg.get("KEEP_CHECKPOINT_EVERY_N_HOURS", 10000.0))
            t_save = time.time()
    if real is not None:
        real.close()                                              # stop the input worker processes
    tr.fold_skipped()
    if rank == 0:
        print("saved", CK.save(args.logdir, tr, cfg.MAX_TO_KEEP, cfg.get("KEEP_CHECKPOINT_EVERY_N_HOURS", 10000.0)))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
