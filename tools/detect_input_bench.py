"""Patches/s of the detection input (detect.py:134-292) on JPEG records: host numpy path vs GPU patch extraction.
usage: python tools/detect_input_bench.py [n_images]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import __graft_entry__ as g
    g.build()
    from multibox_amd import inputs as I
    from multibox_amd.augment import PatchExtractor
    from multibox_amd.config import Cfg
    from tests.test_inputs_cpu import _make_records
    n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    tmp = os.environ.get("TMPDIR", "/tmp")
    path = os.path.join(tmp, "detbench_%d.tfrecords" % n_images)
    if not os.path.exists(path):
        _make_records(path, [(480, 640, []) for _ in range(n_images)])
    cfg = Cfg(dict(INPUT_SIZE=299, DETECTION=dict(
        USE_ORIGINAL_IMAGE=True, ORIGINAL_IMAGE_MAX_TO_KEEP=200, USE_FLIPPED_ORIGINAL_IMAGE=True, FLIPPED_IMAGE_MAX_TO_KEEP=100,
        CROPS=[dict(HEIGHT=299, WIDTH=299, HEIGHT_STRIDE=113, WIDTH_STRIDE=113, FLIP=False, MAX_TO_KEEP=50)])))
    B = 64
    ex = PatchExtractor(B, 299)
    for mode in ("device16", "device4", "device1", "host"):
        t = time.time()
        n = 0
        for batch in I.detect_batches([path] * (6 if mode != "host" else 1), cfg, B, device_patches=(mode != "host"),
                                       decode_threads=int(mode[6:] or 1) if mode != "host" else None):
            if mode != "host":
                x = ex(batch["sources"], batch["patches"])
            else:
                x = torch.from_numpy(batch["images"]).cuda()
            n += B
        torch.cuda.synchronize()
        dt = time.time() - t
        print("%s detection input: %.0f patches/s (%d patches, 10 per 480x640 image)" % (mode, n / dt, n), flush=True)


if __name__ == "__main__":
    main()
