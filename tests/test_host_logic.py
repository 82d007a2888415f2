"""Host-side logic that needs no GPU: config keys, CLI flags, LR schedule, DP segments, patches."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EXAMPLE = """
RANDOM_SEED : 1.0
SESSION_CONFIG : {
  PER_PROCESS_GPU_MEMORY_FRACTION : 0.95
}
NUM_BBOXES_PER_CELL : 7
MAX_NUM_BBOXES : 13
LOCATION_LOSS_ALPHA : 1000.0
BATCH_SIZE : 32
INPUT_SIZE : 299
DETECTION :
  USE_ORIGINAL_IMAGE : true
  ORIGINAL_IMAGE_MAX_TO_KEEP : 200
  CROPS :
    - HEIGHT : 299
      WIDTH : 299
      HEIGHT_STRIDE : 113
      WIDTH_STRIDE : 113
      FLIP : false
      MAX_TO_KEEP : 50
NUM_TRAIN_EXAMPLES : 56945
INITIAL_LEARNING_RATE : 0.01
NUM_EPOCHS_PER_DELAY : 4
LEARNING_RATE_DECAY_FACTOR : 0.94
LEARNING_RATE_STAIRCASE : true
RMSPROP_EPSILON : 1.0
"""


def test_config_keys(tmp_path):
    from multibox_amd.config import parse_config_file, with_defaults
    f = tmp_path / "config.yaml"
    f.write_text(EXAMPLE)
    cfg = with_defaults(parse_config_file(str(f)))
    assert cfg.NUM_BBOXES_PER_CELL == 7 and cfg.RANDOM_SEED == 1.0          # config.yaml.example:1 is a float
    assert cfg.DETECTION.CROPS[0].MAX_TO_KEEP == 50 and cfg.SESSION_CONFIG.PER_PROCESS_GPU_MEMORY_FRACTION == 0.95
    assert cfg.MOVING_AVERAGE_DECAY == 0.9999                               # default filled in
    cfg.BATCH_SIZE = 8                                                      # train.py:362-366 overrides
    assert cfg["BATCH_SIZE"] == 8


@pytest.mark.parametrize("script,flags", [
    ("train.py", ["--tfrecords", "--priors", "--logdir", "--config", "--pretrained_model", "--fine_tune",
                  "--trainable_scopes", "--use_moving_averages", "--restore_moving_averages", "--max_number_of_steps", "--batch_size"]),
    ("detect.py", ["--tfrecords", "--priors", "--checkpoint_path", "--config", "--max_iterations", "--max_detections", "--save_dir"]),
])
def test_cli_flags_match_reference(script, flags):
    """train.py:304-346 / detect.py:466-492."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, script), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for f in flags:
        assert f in out.stdout, f


def test_schedule_matches_oracle():
    from multibox_amd import trainer as T
    from oracle import ref_numpy as R
    assert T.decay_steps(56945, 32, 4) == R.decay_steps(56945, 32, 4) == 7116
    for step in (0, 1, 7115, 7116, 20000):
        assert np.isclose(T.learning_rate(step, 0.01, 7116, 0.94), float(R.learning_rate(step, 0.01, 7116, 0.94)), rtol=1e-6)


def test_extract_patches_matches_oracle():
    from multibox_amd import detect as D
    from oracle import ref_numpy as R
    img = np.random.RandomState(0).rand(480, 640, 3).astype(np.float32)
    patches, offs, res, n = D.extract_patches(img, (299, 299), (113, 113))
    ro, rr = R.extract_patch_offsets((480, 640), (299, 299), (113, 113))
    assert n == 8 and np.array_equal(offs, ro) and np.array_equal(res, rr)
    assert np.array_equal(patches[5], img[113:412, 113:412])
    p2, o2, r2, n2 = D.extract_patches(np.zeros((100, 100, 3), np.float32), (299, 299), (113, 113))
    assert n2 == 0 and p2.shape == (0, 299, 299, 3)                          # detect.py:64-70


def test_backward_segments_cover_gradient_buffer():
    """The data-parallel buckets (one per backward segment) tile the trainable range of Wg exactly."""
    import __graft_entry__ as g
    g.build()
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    for fine_tune in (False, True):
        net = Net(batch=1, input_size=299, k=5, mode="train", fine_tune=fine_tune, device="cpu")
        tr = Trainer.__new__(Trainer)
        tr.net, tr.w_lo = net, (net.head_w_start if fine_tune else 0)
        for n, tail in ((4, 0), (6, 2_000_000)):
            segs = tr._make_segments(n, tail_params=tail)
            assert len(segs) <= n + (1 if tail else 0) and (fine_tune or len(segs) == n + (1 if tail else 0))
            assert sum(len(s[0]) for s in segs) == len(net.bwd_launches)
            hi = net.nW
            for fns, lo, h in segs:
                assert h == hi and lo < h
                hi = lo
            assert hi == tr.w_lo
            if tail and not fine_tune:      # the data-parallel form: the bucket no backward launch overlaps is the small one
                assert segs[-1][2] - segs[-1][1] <= tail and all(s_[2] - s_[1] > tail for s_ in segs[:-1])


def test_generate_aspect_ratios_matches_reference_golden():
    """SURVEY F4 (priors.py:11-183): golden = the reference's own function run on this dataset (tools/gen_golden.py);
    well separated clusters, so the unseeded KMeans of the reference has a unique answer."""
    import os
    from multibox_amd.priors import generate_aspect_ratios
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "aspect_ratios.npz"))
    dataset = [{"id": i, "width": int(w), "height": int(h),
                "object": {"bbox": {"xmin": [b[0]], "ymin": [b[1]], "xmax": [b[2]], "ymax": [b[3]]}}}
               for i, (w, h, b) in enumerate(zip(g["width"], g["height"], g["bbox"]))]
    for warp, key in ((True, "expected_warp"), (False, "expected_nowarp")):
        out = generate_aspect_ratios(dataset, num_aspect_ratios=4, warp_bboxes=warp, random_state=0)
        assert out.shape == (4,)
        assert np.allclose(out, g[key], rtol=1e-6), (out, g[key])      # same order: membership count, largest first
    # the warped clusters are the ones the dataset was built from
    assert np.allclose(sorted(g["expected_warp"]), [0.5, 1.0, 2.0, 3.5], rtol=0.01)


def test_trainable_scopes_ranges():
    """--trainable_scopes (train.py:152-171): variables are selected by re.match of the scope on their name; the
    optimiser ranges cover the whole trainable part of the flat buffers exactly once."""
    import __graft_entry__ as g
    g.build()
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer
    net = Net(batch=1, input_size=299, k=5, mode="train", device="cpu")
    tr = Trainer.__new__(Trainer)
    tr.net, tr.w_lo, tr.bt_lo = net, 0, 0
    assert tr._optimizer_ranges(None) == [("W", 0, net.nW, 1), ("Bt", 0, net.nBt, 1)]
    r = tr._optimizer_ranges(["Multibox", "InceptionResnetV2/Repeat_2/block8_9"])
    for buf, n in (("W", net.nW), ("Bt", net.nBt)):
        rr = [x for x in r if x[0] == buf]
        assert rr[0][1] == 0 and rr[-1][2] == n and all(a[2] == b[1] for a, b in zip(rr, rr[1:]))
        assert all(a[3] != b[3] for a, b in zip(rr, rr[1:]))                       # merged
    names = set(tr.trainable_names)
    assert all(n.startswith("Multibox") or n.startswith("InceptionResnetV2/Repeat_2/block8_9") for n in names)
    assert "Multibox/8x8/Conv/weights" in names and "InceptionResnetV2/Repeat_2/block8_9/Conv2d_1x1/biases" in names
    assert "Multibox/8x8/Conv/BatchNorm/moving_mean" not in names                 # not a trainable variable
    # every selected variable lies in an 'on' range, every other one in an 'off' range
    for name, (buf, off, shape, cpad) in net.param_index.items():
        if buf not in ("W", "Bt"):
            continue
        on = [x[3] for x in r if x[0] == buf and x[1] <= off < x[2]]
        assert on == [int(name in names)], name
    # --fine_tune restricts the candidates to the detection heads first (train.py:229-232)
    tr.w_lo, tr.bt_lo = net.head_w_start, net.head_bt_start
    r = tr._optimizer_ranges(["InceptionResnetV2"])
    assert all(x[3] == 0 for x in r) and tr.trainable_names == []


def test_oracle_nms_known_answers():
    """oracle.ref_numpy.nms_greedy (row N1; the reference has no NMS, so these are hand-computed answers)."""
    from oracle import ref_numpy as R
    import numpy as np
    b = np.array([[0, 0, 1, 1], [0, 0, 1, .5], [.5, .5, 1.5, 1.5], [2, 2, 3, 3], [0, 0, 1, 1]], np.float64)
    # IoU(0,1) = .5, IoU(0,2) = .25/1.75, IoU(0,4) = 1, IoU(1,2) = 0 (touching edge), box 3 is disjoint
    assert R.nms_greedy(b, 0.5).tolist() == [0, 1, 2, 3]              # strictly greater than the threshold suppresses
    assert R.nms_greedy(b, 0.49).tolist() == [0, 2, 3]
    assert R.nms_greedy(b, 0.1).tolist() == [0, 3]
    assert R.nms_greedy(b, 1.0).tolist() == [0, 1, 2, 3, 4]
    assert R.nms_greedy(np.zeros((0, 4)), 0.5).tolist() == []


def test_checkpoint_retention_like_tf_saver(tmp_path):
    """tf.train.Saver(max_to_keep, keep_checkpoint_every_n_hours) (train.py:282-286): the newest max_to_keep files stay, and a
    file about to be deleted is kept for good if it is at least N hours younger than the previous permanent one."""
    import json
    import time
    from multibox_amd import checkpoint as CK
    d = str(tmp_path)
    t0 = time.time() - 10 * 3600
    for i, hours in enumerate([0.0, 0.5, 1.0, 2.5, 3.0, 5.2, 5.3, 6.0]):          # "saved" at t0 + hours
        p = os.path.join(d, "model.ckpt-%d.pt" % (100 * i))
        open(p, "wb").write(b"x")
        os.utime(p, (t0 + 3600 * hours, t0 + 3600 * hours))
        CK._prune(d, max_to_keep=2, keep_every_n_hours=2.0, now=t0)
    left = sorted(os.listdir(d))
    # permanent: 300 (2.5 h > first deadline t0 + 2 h; the deadline moves to 4 h) and 500 (5.2 h > 4 h; -> 6 h);
    # newest two: 600, 700; everything else was deleted when it fell out of the window
    assert [f for f in left if f.endswith(".pt")] == ["model.ckpt-300.pt", "model.ckpt-500.pt", "model.ckpt-600.pt", "model.ckpt-700.pt"]
    assert json.load(open(os.path.join(d, "checkpoint_retention.json")))["kept"] == ["model.ckpt-300.pt", "model.ckpt-500.pt"]
    assert CK.latest_checkpoint(d).endswith("model.ckpt-700.pt")


def test_record_text_is_byte_identical_to_json_dump_of_the_dicts():
    """detect.py:438-460 writes json.dump(list of {"image_id", "bbox", "score"}); multibox_amd/records.py produces the same
    TEXT without building the dicts (in worker processes) -- byte for byte, including non-finite numbers, -0.0, tiny and
    huge values, string ids that need escaping, empty patches and empty batches."""
    import json
    from multibox_amd import detect as D, records as REC
    rng = np.random.RandomState(0)
    B, K = 16, 40
    boxes = rng.rand(B, K, 4) * 300
    boxes[3, 5, 2] = 1e-7; boxes[4, 0, 0] = 1e22; boxes[5, 1, 1] = float("nan"); boxes[6, 0, 0] = float("inf"); boxes[7, 0, 0] = -0.0
    scores = rng.rand(B, K).astype(np.float32)
    scores[2, 3] = np.float32("nan"); scores[1, 0] = np.float32(1e-30)
    count = np.array([40, 25, 0] + [7] * 13, np.int32)
    ids = list(range(B - 1)) + ['abc"x\\']
    want = json.dumps(D.results_to_json_records(boxes, scores, count, ids))
    recs = REC.results_to_json_text(boxes, scores, count, ids)
    assert len(recs) == int(count.sum()) and REC.records_to_json(recs) == want
    n, chunk = REC.batch_chunk(boxes, scores, count, ids)
    assert n == int(count.sum()) and REC.records_to_json([chunk]) == want
    # chunks of several batches, empty ones in between, and no records at all
    n2, chunk2 = REC.batch_chunk(boxes[:2], scores[:2], count[:2], ids[:2])
    assert REC.records_to_json([chunk2, "", chunk]) == json.dumps(
        D.results_to_json_records(boxes[:2], scores[:2], count[:2], ids[:2]) + D.results_to_json_records(boxes, scores, count, ids))
    assert REC.records_to_json([]) == json.dumps([]) == "[]"
    assert REC.records_to_json([""]) == "[]"


def test_shipped_tile_table_is_well_formed():
    """multibox_amd/tune_cache.json (the measured tile choice per layer shape, tools/tune_by_trace.py): every key parses,
    every value is a configuration the library knows for that kind of launch, and the persistent kernels are named only
    for shapes they cover (igemm7: pointwise, unit stride, 64 < K <= 384, no statistics, at most 32 column tiles; igemm5: no
    stride-2 data gradient, no float32 head epilogue)."""
    import ast
    import json
    import os
    from multibox_amd import ops
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multibox_amd", "tune_cache.json")
    table = json.load(open(path))
    assert len(table) > 300
    n7 = n5 = 0
    for key, cfg in table.items():
        i3 = key.endswith("#i3")
        k = ast.literal_eval(key[:-3] if i3 else key)
        what = k[0]
        assert what in ("fwd", "dgrad", "wgrad") and isinstance(cfg, int), key
        if what == "wgrad":
            assert 0 <= cfg <= 10 and not i3, key
            continue
        if len(k) == 19:                 # a residual launch that also writes the relu sign bits (engine._tune)
            assert k[18] == "bits" and k[13] == ops.EPI_RESIDUAL and k[17] == 1, key
            k = k[:18]
        (_, N, H, W, Cin, K, R, S, stride, pt, pl, c_out, c_in, epilogue, stats, accumulate, skip, relu) = k
        if i3:
            assert 0 <= cfg <= ops.N_TILE_CONFIGS, key
            continue
        assert 0 <= cfg <= ops.N_TILE_CONFIGS or cfg in ops.I5_TILE_CONFIGS or cfg == ops.I7_TILE_CONFIG, key
        if cfg > ops.I5_FLAG:
            assert epilogue != ops.EPI_STORE_F32 and not (what == "dgrad" and stride == 2), key
        if cfg in ops.I5_TILE_CONFIGS:
            n5 += 1
        if cfg == ops.I7_TILE_CONFIG:
            n7 += 1
            assert R == S == 1 and stride == 1 and pt == pl == 0 and not stats and 64 < c_in <= 384, key
            assert (c_out + 127) // 128 <= ops.I7_COUNTERS, key
    assert n5 > 50 and n7 >= 1


def test_patch_meta_packing_matches_the_c_struct():
    """detect.make_patch_meta (whole numpy columns) produces byte for byte the array of mbx_patch_meta structs that the
    ctypes declaration of include/mbx.h lays out."""
    import ctypes
    from multibox_amd import detect as D, _lib
    B = 7
    rng = np.random.RandomState(0)
    off, dm, hw = rng.randint(0, 500, (B, 2)), rng.randint(1, 500, (B, 2)), rng.randint(1, 900, (B, 2))
    fl, k, rs = rng.randint(0, 2, (B, 1)), rng.randint(1, 200, (B, 1)), rng.rand(B, 4).astype(np.float32)
    got = D.make_patch_meta(off, dm, fl, rs, k, hw, device="cpu").numpy()
    arr = (_lib.PatchMeta * B)()
    for b in range(B):
        m = arr[b]
        m.offset_y, m.offset_x, m.patch_h, m.patch_w = int(off[b][0]), int(off[b][1]), int(dm[b][0]), int(dm[b][1])
        m.image_h, m.image_w, m.is_flipped, m.max_to_keep = int(hw[b][0]), int(hw[b][1]), int(fl[b][0]), int(k[b][0])
        for i in range(4):
            m.restrictions[i] = float(rs[b][i])
    want = np.frombuffer(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)), dtype=np.uint8)
    assert got.shape == want.shape and bool((got == want).all())


def test_bench_gpus_flag_starts_that_many_ranks_or_refuses():
    """`bench.py --gpus N`: N > 1 without a rendezvous in the environment -> ONE child
    `python -m torch.distributed.run --nproc-per-node N ... bench.py <same argv>` on 127.0.0.1 (spawned, never exec'ed);
    fewer visible devices than N, or a launcher's WORLD_SIZE that disagrees with --gpus -> refusal (exit code 2), never an
    `n_gpus: 1` line under a larger request.  --gpus 1 and the driver's own torchrun form run as they always did."""
    import bench
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert bench.launch_plan(1, {}, ["--gpus", "1"], 0) == ("run", None)
    assert bench.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}, argv, 0) == ("run", None)        # the driver's torchrun form
    act, cmd = bench.launch_plan(8, {}, argv, 8, script="/x/bench.py", port=29999)
    assert act == "spawn"
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[cmd.index("/x/bench.py") + 1:] == argv                                               # same arguments for every rank
    act, cmd = bench.launch_plan(2, {}, ["--gpus", "2"], 4)                                         # a free port is picked
    assert act == "spawn" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    act, msg = bench.launch_plan(8, {}, argv, 1)
    assert act == "refuse" and "only 1 GPU" in msg
    act, msg = bench.launch_plan(8, {"WORLD_SIZE": "2"}, argv, 8)
    assert act == "refuse" and "WORLD_SIZE=2" in msg
    act, msg = bench.launch_plan(1, {"WORLD_SIZE": "4"}, ["--gpus", "1"], 8)
    assert act == "refuse"
    assert bench.launch_plan(0, {}, [], 8)[0] == "refuse"
    # end to end on this GPU-less container: the refusal is an exit code and a message, and nothing JSON-like on stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "--gpus 2" in r.stderr and "{" not in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-500:])


def test_grid_barrier_cap_is_derived_from_rccl_channels():
    """the workgroup cap of the grid-barrier launches in a data-parallel run = CUs - RCCL's channels."""
    from multibox_amd.dist import bn_max_workgroups_for, rccl_cu_reserve
    assert bn_max_workgroups_for(1, 256, {})[0] == 0
    assert bn_max_workgroups_for(8, 256, {}) == (192, {"reserve": 64, "source": "RCCL default channel ceiling (64)"})
    assert bn_max_workgroups_for(8, 256, {"NCCL_MAX_NCHANNELS": "32"})[0] == 224
    assert rccl_cu_reserve(256, {"NCCL_MAX_NCHANNELS": "junk"})[0] == 64
    assert rccl_cu_reserve(256, {"NCCL_MAX_NCHANNELS": "512"})[0] == 128         # never more than half the chip
