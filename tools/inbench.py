"""Throughput of the training input path on real JPEG records: worker processes -> ring -> prefetcher (-> GPU
augmentation), without a training step.  usage: python tools/inbench.py [workers] [host|device] [n_batches]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from multibox_amd.config import Cfg
    from multibox_amd.input_workers import ParallelTrainInput, DevicePrefetcher
    from tests.test_inputs_cpu import _make_records
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    mode = sys.argv[2] if len(sys.argv) > 2 else "device"
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    tmp = os.environ.get("TMPDIR", "/tmp")
    path = os.path.join(tmp, "inbench.tfrecords")
    if not os.path.exists(path):
        _make_records(path, [(480, 640, [[.1, .2, .5, .6], [.3, .3, .9, .8]][: i % 3]) for i in range(256)])
    cfg = Cfg(dict(INPUT_SIZE=299, DO_RANDOM_FLIP_LEFT_RIGHT=True, DO_COLOR_DISTORTION=0.5, COLOR_DISTORT_FAST=False,
                   DO_RANDOM_CROP=0.5, RANDOM_CROP_MIN_OBJECT_COVERED=0.7, RANDOM_CROP_ASPECT_RATIO_RANGE=[0.7, 1.4],
                   RANDOM_CROP_AREA_RANGE=[0.5, 1.0], RANDOM_CROP_MAX_ATTEMPTS=100, RANDOM_CROP_MINIMUM_AREA=50,
                   DO_RANDOM_BBOX_SHIFT=0.5, RANDOM_BBOX_SHIFT_EXTENT=4))
    B = 64
    src = ParallelTrainInput([path], cfg, B, 5, num_workers=workers, seed=1, shuffle=True, device_augment=(mode == "device"))
    pre = DevicePrefetcher(src, B, 299, 5, device="cuda", depth=2)
    for _ in range(4):
        pre.next()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(nb):
        images, bb, n, ids = pre.next()
    torch.cuda.synchronize()
    dt = time.time() - t
    print("%s augmentation, %d workers: %.0f images/s (%d batches of %d, 480x640 JPEGs, full colour ordering)" % (
        mode, workers, nb * B / dt, nb, B), flush=True)
    pre.close()


if __name__ == "__main__":
    main()
