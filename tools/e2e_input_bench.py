"""train.py on real JPEG records at BATCH_SIZE 64, with the augmentation on the GPU and on the host: the images/s the
training log reports once the pipeline is warm.  usage: python tools/e2e_input_bench.py [workers] [steps]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = """
NUM_BBOXES_PER_CELL : 5
MAX_NUM_BBOXES : 13
LOCATION_LOSS_ALPHA : 1000.0
BATCH_SIZE : 64
INPUT_SIZE : 299
NUM_TRAIN_EXAMPLES : 56945
NUM_TRAIN_ITERATIONS : 1000000
LOG_EVERY_N_STEPS : 50
SAVE_INTERVAL_SECS : 100000
NUM_INPUT_THREADS : %d
INPUT_AUGMENT_ON_DEVICE : %s
INPUT_AUGMENT_KERNELS_ON_SIDE_STREAM : %s
QUEUE_CAPACITY : 1000
QUEUE_MIN : 96
DO_RANDOM_FLIP_LEFT_RIGHT : true
DO_RANDOM_BBOX_SHIFT : 0.5
RANDOM_BBOX_SHIFT_EXTENT : 4
DO_RANDOM_CROP : 0.5
RANDOM_CROP_MIN_OBJECT_COVERED : 0.7
RANDOM_CROP_ASPECT_RATIO_RANGE : [0.7, 1.4]
RANDOM_CROP_AREA_RANGE : [0.5, 1.0]
RANDOM_CROP_MAX_ATTEMPTS : 100
RANDOM_CROP_MINIMUM_AREA : 50
DO_COLOR_DISTORTION : 0.5
COLOR_DISTORT_FAST : false
DETECTION :
  USE_ORIGINAL_IMAGE : true
  ORIGINAL_IMAGE_MAX_TO_KEEP : 200
"""


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    from multibox_amd import priors as PR
    from tests.test_inputs_cpu import _make_records
    tmp = os.environ.get("TMPDIR", "/tmp")
    rec = os.path.join(tmp, "e2e.tfrecords")
    if not os.path.exists(rec):
        _make_records(rec, [(480, 640, [[.1, .2, .5, .6], [.3, .3, .9, .8]][: i % 3]) for i in range(256)])
    pri = os.path.join(tmp, "e2e_priors.pkl")
    PR.save_priors(pri, PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]))
    for on_device, side in (("true", "true"), ("true", "false"), ("false", "true")):
        cfg = os.path.join(tmp, "e2e_%s.yaml" % on_device)
        open(cfg, "w").write(CFG % (workers, on_device, side))
        logdir = os.path.join(tmp, "e2e_log_%s" % on_device)
        subprocess.run(["rm", "-rf", logdir])
        n = steps if on_device == "true" else min(max(steps // 3, 100), 600)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--priors", pri, "--logdir", logdir,
                            "--config", cfg, "--tfrecords", rec, "--max_number_of_steps", str(n)],
                           capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
        if r.returncode != 0:
            print(r.stdout[-1500:], r.stderr[-1500:])
            raise SystemExit(1)
        recs = [json.loads(l) for l in open(os.path.join(logdir, "train_log.jsonl"))]
        import math
        print("augmentation on the %s%s, %d workers: images/s per 50-step window = %s; non-finite total_loss in %d of %d windows" % (
            "GPU" if on_device == "true" else "host", (" (kernels on the %s stream)" % ("side" if side == "true" else "training")) if on_device == "true" else "",
            workers, [round(x["images_per_sec"]) for x in recs][:8] + ['...'] + [round(x["images_per_sec"]) for x in recs][-2:], sum(not math.isfinite(x["total_loss"]) for x in recs), len(recs)), flush=True)


if __name__ == "__main__":
    main()
