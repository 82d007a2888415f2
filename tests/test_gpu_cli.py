"""End-to-end drop-in CLIs on the GPU: train.py (synthetic input) -> checkpoint -> detect.py -> results JSON
with the reference's record format (detect.py:438-460)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG = """
NUM_BBOXES_PER_CELL : 5
MAX_NUM_BBOXES : 13
LOCATION_LOSS_ALPHA : 1000.0
BATCH_SIZE : 4
INPUT_SIZE : 299
NUM_TRAIN_EXAMPLES : 56945
NUM_TRAIN_ITERATIONS : 1000000
LOG_EVERY_N_STEPS : 1
BATCHNORM_MOVING_AVERAGE_DECAY : 0.3
INITIAL_LEARNING_RATE : 0.00001
DETECTION :
  USE_ORIGINAL_IMAGE : true
  ORIGINAL_IMAGE_MAX_TO_KEEP : 200
"""


def test_train_then_detect(tmp_path):
    import __graft_entry__ as g
    g.build()
    from multibox_amd import priors as PR
    cfg = tmp_path / "config.yaml"
    cfg.write_text(CFG)
    pri = tmp_path / "priors.pkl"
    PR.save_priors(str(pri), PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]))
    logdir, outdir = tmp_path / "log", tmp_path / "out"
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--priors", str(pri), "--logdir", str(logdir),
                        "--config", str(cfg), "--max_number_of_steps", "3", "--synthetic"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(logdir / "model.ckpt-3.pt")
    log = [json.loads(l) for l in open(logdir / "train_log.jsonl")]
    assert len(log) == 3 and all(np.isfinite(x["total_loss"]) for x in log)
    assert abs(log[0]["total_loss"] - (log[0]["location_loss"] + log[0]["confidence_loss"])) < 0.01 * log[0]["total_loss"] + 5
    # resume: slim.learning.train picks the latest checkpoint of logdir (train.py:33-37)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--priors", str(pri), "--logdir", str(logdir),
                        "--config", str(cfg), "--max_number_of_steps", "4", "--synthetic", "--fine_tune"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "Resumed from" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(logdir / "model.ckpt-4.pt")
    # The fine-tune step runs the backbone with FROZEN batch norm, i.e. on the moving statistics of the three steps above.
    # With the reference's decay (0.9997) those are still (0, 1) after three steps, nothing is normalised through the 40
    # residual blocks and the step's location loss comes out at 1e32 .. inf (measured: tests/debug/cli_finetune_probe.sh) --
    # finite or not by luck of the three chaotic steps before it, which made this test fail once in a while ("bipartite
    # matching failed: non-finite predictions").  BATCHNORM_MOVING_AVERAGE_DECAY 0.3 in the config above lets the
    # statistics follow within three steps, and INITIAL_LEARNING_RATE 1e-5 keeps the weights those statistics were
    # taken on (at 0.01 three RMSProp steps from a random start move them enough for 1e18 again); the step must then
    # be an ordinary one:
    log = [json.loads(l) for l in open(logdir / "train_log.jsonl")]
    assert len(log) == 4 and np.isfinite(log[3]["total_loss"]) and log[3]["total_loss"] < 1e6, log[3]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "detect.py"), "--priors", str(pri), "--checkpoint_path", str(logdir),
                        "--config", str(cfg), "--save_dir", str(outdir), "--synthetic", "8", "--max_iterations", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.load(open(outdir / "results-dense-4.json"))          # global step parsed from the checkpoint name
    assert len(res) > 0 and set(res[0]) == {"image_id", "bbox", "score"}
    b = np.array([x["bbox"] for x in res])
    assert b.shape[1] == 4 and (b >= 0).all() and (b <= 1).all()       # clipped like detect.py:413; no ordering is enforced
    per_image = {}
    for x in res:
        per_image.setdefault(x["image_id"], []).append(x["score"])
    assert len(per_image) == 8 and all(len(v) <= 200 for v in per_image.values())
    assert all(v == sorted(v, reverse=True) for v in per_image.values())    # detect.py:423 sort by confidence
    # the same checkpoint as a TensorFlow V1 table (slim names, HWIO filters, EMA shadows) gives the same
    # detections through detect.py (detect.py:336-346 restores the shadows), and initialises train.py
    # --pretrained_model --fine_tune (train.py:15-90)
    tfck = tmp_path / "tf" / "model.ckpt-4"
    os.makedirs(tmp_path / "tf")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_checkpoint.py"), str(logdir), str(tfck)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    (tmp_path / "tf" / "checkpoint").write_text('model_checkpoint_path: "model.ckpt-4"\nall_model_checkpoint_paths: "model.ckpt-4"\n')
    out2 = tmp_path / "out_tf"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "detect.py"), "--priors", str(pri), "--checkpoint_path", str(tmp_path / "tf"),
                        "--config", str(cfg), "--save_dir", str(out2), "--synthetic", "8", "--max_iterations", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert json.load(open(out2 / "results-dense-4.json")) == res
    log2 = tmp_path / "log_ft"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--priors", str(pri), "--logdir", str(log2),
                        "--config", str(cfg), "--max_number_of_steps", "1", "--synthetic", "--fine_tune",
                        "--pretrained_model", str(tfck), "--use_moving_averages", "--restore_moving_averages"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "Initialised from" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(log2 / "model.ckpt-1.pt")


def test_finetune_after_fresh_start_is_handled_cleanly(tmp_path):
    """the SAME two runs with the reference's own BATCHNORM_MOVING_AVERAGE_DECAY (0.9997) and learning rate
    (0.01, train.py defaults): three steps from a random start, then `--fine_tune` on moving statistics that are still (0, 1) --
    activations of ~1e16 reach the training-mode head batch norms (outside the fixed-point range of the statistics rows) and
    the step's losses are 1e32 .. inf.  Whatever the three chaotic steps made of it, the run must end in ONE of two clean
    ways -- a logged step whose loss is a float (huge or inf, never a NaN disguised as a finite number), or the reference's
    own failure mode, the matcher's error (loss.py:82: the py_func raises on non-finite predictions) -- and a checkpoint
    that an inference run can still read.  Never a hang, a crash without message, or finite garbage."""
    import __graft_entry__ as g
    g.build()
    from multibox_amd import priors as PR
    cfg = tmp_path / "config.yaml"
    cfg.write_text(CFG.replace("BATCHNORM_MOVING_AVERAGE_DECAY : 0.3", "BATCHNORM_MOVING_AVERAGE_DECAY : 0.9997")
                   .replace("INITIAL_LEARNING_RATE : 0.00001", "INITIAL_LEARNING_RATE : 0.01"))
    pri = tmp_path / "priors.pkl"
    PR.save_priors(str(pri), PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]))
    logdir = tmp_path / "log"
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--priors", str(pri), "--logdir", str(logdir),
                        "--config", str(cfg), "--max_number_of_steps", "3", "--synthetic"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--priors", str(pri), "--logdir", str(logdir),
                        "--config", str(cfg), "--max_number_of_steps", "4", "--synthetic", "--fine_tune"],
                       capture_output=True, text=True, timeout=900, env=env)
    log = [json.loads(l) for l in open(logdir / "train_log.jsonl") if "total_loss" in l]
    if r.returncode == 0:
        assert len(log) == 4 and isinstance(log[3]["total_loss"], float), log[-1]
        assert not np.isnan(log[3]["location_loss"]) or not np.isfinite(log[3]["total_loss"]), log[3]
    else:
        assert "bipartite matching failed" in (r.stdout + r.stderr), r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(logdir / "model.ckpt-3.pt")


@pytest.mark.parametrize("on_device", ["true", "false"])
def test_train_from_tfrecords_with_input_workers(tmp_path, on_device):
    """train.py --tfrecords: JPEG records -> NUM_INPUT_THREADS worker processes (multibox_amd/input_workers.py) ->
    pinned buffers -> asynchronous H2D (-> mbx_augment_batch when INPUT_AUGMENT_ON_DEVICE) -> Trainer, six steps with
    every augmentation switched on."""
    import __graft_entry__ as g
    g.build()
    from multibox_amd import priors as PR
    from tests.test_inputs_cpu import _make_records
    cfg = tmp_path / "config.yaml"
    cfg.write_text(CFG.replace("DETECTION :", """NUM_INPUT_THREADS : 3
INPUT_AUGMENT_ON_DEVICE : ON_DEVICE
QUEUE_CAPACITY : 24
QUEUE_MIN : 8
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
DETECTION :""".replace("ON_DEVICE", on_device)))
    pri = tmp_path / "priors.pkl"
    PR.save_priors(str(pri), PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.]))
    rec = str(tmp_path / "train.tfrecords")
    _make_records(rec, [(300 + 11 * i, 330 + 7 * i, [[.1, .2, .5, .6], [.3, .3, .9, .8]][: i % 3]) for i in range(12)])
    logdir = tmp_path / "log"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--priors", str(pri), "--logdir", str(logdir),
                        "--config", str(cfg), "--tfrecords", rec, "--max_number_of_steps", "6"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    recs = [json.loads(l) for l in open(logdir / "train_log.jsonl")]
    assert [x["global_step"] for x in recs] == [1, 2, 3, 4, 5, 6]
    assert all(np.isfinite(x["total_loss"]) and x["location_loss"] >= 0 and x["confidence_loss"] > 0 for x in recs)
    assert any(x["location_loss"] > 0 for x in recs)             # the boxes arrived (two of three images have some)
    assert "saved" in r.stdout


@pytest.mark.parametrize("on_device", ["true", "false"])
def test_detect_from_tfrecords(tmp_path, on_device):
    """detect.py --tfrecords: the reference's multi-crop input (detect.py:134-292) end to end on three JPEG images, the
    patches cut and resized on the GPU (mbx_extract_patches) or on the host."""
    import __graft_entry__ as g
    g.build()
    import torch
    from multibox_amd import priors as PR, checkpoint as CK
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    from tests.test_inputs_cpu import _make_records
    cfg = tmp_path / "config.yaml"
    cfg.write_text(CFG.replace("DETECTION :", "INPUT_AUGMENT_ON_DEVICE : %s\nDETECTION :" % on_device) + """  USE_FLIPPED_ORIGINAL_IMAGE : true
  FLIPPED_IMAGE_MAX_TO_KEEP : 100
  CROPS :
    - HEIGHT : 299
      WIDTH : 299
      HEIGHT_STRIDE : 113
      WIDTH_STRIDE : 113
      FLIP : false
      MAX_TO_KEEP : 50
""")
    pri = tmp_path / "priors.pkl"
    priors = PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.])
    PR.save_priors(str(pri), priors)
    rec = str(tmp_path / "val.tfrecords")
    _make_records(rec, [(320, 420, []), (300, 300, []), (412, 412, [])])      # patches: 2+1, 2+1, 2+4 = 12
    net = Net(batch=4, input_size=299, k=5, mode="train")
    tr = Trainer(net, np.array(priors, np.float32), use_graph=False)
    CK.save(str(tmp_path / "log"), tr)
    del tr, net
    torch.cuda.empty_cache()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "detect.py"), "--priors", str(pri), "--checkpoint_path", str(tmp_path / "log"),
                        "--config", str(cfg), "--save_dir", str(tmp_path / "out"), "--tfrecords", rec],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.load(open(tmp_path / "out" / "results-dense-0.json"))
    ids = sorted(set(x["image_id"] for x in res))
    assert ids == [1000, 1001, 1002] and "Step: 3" in r.stdout
    b = np.array([x["bbox"] for x in res])
    assert (b >= 0).all() and (b <= 1).all()


def test_eval_cli(tmp_path):
    """eval.py (eval.py:25-246): inference-mode forward with the EMA weights, top-100 per image, COCO bbox AP/AR summary."""
    import __graft_entry__ as g
    g.build()
    import torch
    from multibox_amd import priors as PR, checkpoint as CK
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    from tests.test_inputs_cpu import _make_records
    cfg = tmp_path / "config.yaml"
    cfg.write_text(CFG)
    pri = tmp_path / "priors.pkl"
    priors = PR.generate_priors([1, 2, 3, 1 / 2., 1 / 3.])
    PR.save_priors(str(pri), priors)
    rec = str(tmp_path / "val.tfrecords")
    _make_records(rec, [(320, 420, [[.1, .1, .6, .7]]), (300, 300, []), (412, 412, [[.2, .3, .9, .8], [.0, .0, .3, .3]]),
                        (299, 299, [[.4, .4, .6, .6]]), (310, 330, [[.1, .1, .2, .2]])])            # 5 images: one batch of 4
    net = Net(batch=4, input_size=299, k=5, mode="train")
    tr = Trainer(net, np.array(priors, np.float32), use_graph=False)
    CK.save(str(tmp_path / "log"), tr)
    del tr, net
    torch.cuda.empty_cache()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "eval.py"), "--priors", str(pri), "--checkpoint_path", str(tmp_path / "log"),
                        "--config", str(cfg), "--summary_dir", str(tmp_path / "sum"), "--tfrecords", rec],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.load(open(tmp_path / "sum" / "eval-0.json"))
    assert out["images"] == 4 and len(out["stats"]) == 12 and len(out["summary"]) == 12
    assert "Average Precision  (AP) @[ IoU=0.50:0.95 | area=   all | maxDets=100 ]" in out["summary"]
    assert all(-1.0 <= v <= 1.0 for v in out["stats"]) and "Step: 1" in r.stdout
    assert out["stats"][0] >= 0.0                      # an untrained net scores ~0, but the metric is defined (4 gt boxes)


def test_bench_contract_over_rccl(tmp_path):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per GPU, RCCL):
    on a one-GPU box MBX_FORCE_DIST=1 keeps the process group, the bucketed all-reduce between the backward graph
    segments and the grid cap of the BN backward in the path.  One JSON line with the contract's keys."""
    env = dict(os.environ, PYTHONPATH=ROOT, MBX_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--batch", "8", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["matching_ok"] and out["grid_barrier_timeouts"] == 0 and out["value"] > 0
    assert out["roofline"]["bound"] == "mfma" and 0 < out["roofline"]["frac"] < 1


def test_bench_gpus_2_on_a_one_gpu_box_refuses():
    """`bench.py --gpus 2` where one device is visible: non-zero exit and a reason, no JSON line labelled with fewer GPUs."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], capture_output=True, text=True,
                       timeout=300, env=dict(env, PYTHONPATH=ROOT))
    assert r.returncode == 2 and "only 1 GPU" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
