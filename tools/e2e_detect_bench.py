"""detect.py on JPEG records end to end (decode -> patches -> forward -> decode/filter/top-K -> JSON records): the
patches/s line it prints, GPU input path and host input path.  usage: python tools/e2e_detect_bench.py [n_images]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = """
NUM_BBOXES_PER_CELL : 5
MAX_NUM_BBOXES : 13
LOCATION_LOSS_ALPHA : 1000.0
BATCH_SIZE : %(batch)s
INPUT_SIZE : 299
NUM_TRAIN_EXAMPLES : 56945
NUM_TRAIN_ITERATIONS : 1000000
NUM_INPUT_THREADS : %(threads)s
NUM_DECODE_PROCESSES : %(procs)s
INPUT_AUGMENT_ON_DEVICE : %(on_device)s
DETECTION :
  USE_ORIGINAL_IMAGE : true
  ORIGINAL_IMAGE_MAX_TO_KEEP : 200
  USE_FLIPPED_ORIGINAL_IMAGE : true
  FLIPPED_IMAGE_MAX_TO_KEEP : 100
  CROPS :
    - HEIGHT : 299
      WIDTH : 299
      HEIGHT_STRIDE : 113
      WIDTH_STRIDE : 113
      FLIP : false
      MAX_TO_KEEP : 50
"""


def main():
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    from multibox_amd import priors as PR, checkpoint as CK
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    from tests.test_inputs_cpu import _make_records
    tmp = os.environ.get("TMPDIR", "/tmp")
    rec = os.path.join(tmp, "e2e_det_%d.tfrecords" % n_images)
    if not os.path.exists(rec):
        _make_records(rec, [(480, 640, []) for _ in range(n_images)])
    pri = os.path.join(tmp, "e2e_det_priors.pkl")
    priors = PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.])
    PR.save_priors(pri, priors)
    logdir = os.path.join(tmp, "e2e_det_log")
    subprocess.run(["rm", "-rf", logdir])
    net = Net(batch=4, input_size=299, k=5, mode="train")
    tr = Trainer(net, np.array(priors, np.float32), use_graph=False)
    CK.save(logdir, tr)
    del tr, net
    torch.cuda.empty_cache()
    # E2E_BATCH / E2E_THREADS / E2E_PROCS: BATCH_SIZE, NUM_INPUT_THREADS and NUM_DECODE_PROCESSES of the run (64 / 8 / 0); E2E_HOST=1 also times the host input path (~60 s)
    for on_device in (("true", "false") if os.environ.get("E2E_HOST") else ("true",)):
        cfg = os.path.join(tmp, "e2e_det_%s.yaml" % on_device)
        open(cfg, "w").write(CFG % dict(on_device=on_device, batch=os.environ.get("E2E_BATCH", "64"), threads=os.environ.get("E2E_THREADS", "8"), procs=os.environ.get("E2E_PROCS", "0")))
        prof = ["-m", "cProfile", "-s", "tottime"] if os.environ.get("MBX_E2E_PROFILE") else []      # where the host time goes
        r = subprocess.run([sys.executable] + prof + [os.path.join(ROOT, "detect.py"), "--priors", pri, "--checkpoint_path", logdir,
                            "--config", cfg, "--save_dir", os.path.join(tmp, "e2e_det_out"), "--tfrecords", rec],
                           capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
        if r.returncode != 0:
            print(r.stdout[-1500:], r.stderr[-1500:])
            raise SystemExit(1)
        if prof:
            lines = r.stdout.splitlines()
            i = next((k for k, l in enumerate(lines) if "tottime" in l and "ncalls" in l), len(lines))
            print("\n".join(lines[i:i + 28]))
            if on_device == "true":
                print("input on the GPU:", [l for l in lines if "patches/s" in l][-1], flush=True)
                break
        print("input on the %s:" % ("GPU" if on_device == "true" else "host"),
              [l for l in r.stdout.splitlines() if "patches/s" in l][-1], flush=True)


if __name__ == "__main__":
    main()
