"""Training steps with ANOTHER stream's kernels sharing the CUs all the time (the situation of data-parallel runs, where
RCCL's kernels overlap the backward pass, and of the input augmentation on a side stream): every parameter must stay
finite.  Round 2 found the step's LDS-ring kernels reading a ring slot's previous contents about once in 10^5 launches
under exactly this contention (bare s_barrier without compiler fences); tests/test_gpu_assembled.py runs this.
usage: python tools/side_stream_stress.py [steps] [compare|infer] [saturate]      prints one JSON line
  compare   (with MBX_DETERMINISTIC=1) the same steps without and with the noise must leave bit-identical parameters
  saturate  noise = tools/noise.hip instead of the augmentation kernels: a 1 GB streaming copy (HBM), a float-atomics
            storm (memory-side atomic units) and an L2 -> LDS LDS-DMA hammer, 1024 blocks each, looping on the second
            stream with every CU oversubscribed -- the memory system as a 240 MB RCCL all-reduce leaves it"""
import json
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def saturating_noise(torch):
    """Returns launch(): one round of the three tools/noise.hip kernels on the CURRENT stream."""
    import ctypes as C
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "_bin", "libnoise.so")
    src = os.path.join(here, "noise.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, src])
    lib = C.CDLL(so)
    lib.noise_launch.restype = C.c_int
    lib.noise_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p,
                                 C.c_int, C.c_int, C.c_void_p]
    nbytes = 1 << 30
    a = torch.zeros(nbytes // 4, dtype=torch.int32, device="cuda")
    b = torch.zeros(nbytes // 4, dtype=torch.int32, device="cuda")
    table = torch.zeros(1 << 24, dtype=torch.float32, device="cuda")          # 64 MB of atomic targets
    l2 = torch.zeros(1 << 19, dtype=torch.int32, device="cuda")               # 2 MB: L2-resident source of the LDS-DMA hammer
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    keep = (a, b, table, l2, sink)

    def launch(_keep=keep):
        rc = lib.noise_launch(a.data_ptr(), b.data_ptr(), nbytes, table.data_ptr(), 1 << 24, l2.data_ptr(), 1 << 21,
                              sink.data_ptr(), 7, 1, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    return launch


def main():
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from multibox_amd import inputs as I, priors as PR
    from multibox_amd.augment import BatchAugmenter
    from multibox_amd.engine import Net
    from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS
    from multibox_amd.trainer import Trainer
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    if len(sys.argv) > 2 and sys.argv[2] == "infer":
        return infer_mode(steps)
    saturate = "saturate" in sys.argv[2:]
    quiet_first = len(sys.argv) > 2 and sys.argv[2] == "compare"   # MBX_DETERMINISTIC=1 ... <steps> compare: the same steps
    #   without and with the noise must leave bit-identical parameters (finite-but-stale reads would show here)
    B, S, k = 64, 299, 5
    net = Net(batch=B, input_size=S, k=k, mode="train", seed=2)
    pri = PR.priors_for_input_size(DEFAULT_ASPECT_RATIOS[k], S).astype(np.float32)
    tr = Trainer(net, pri, max_num_bboxes=13, location_loss_alpha=1000.0)
    images, gt, n = synthetic_batch(B, S, 13, seed=0)
    tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())
    # the noise: the input-augmentation kernels (long fp64 elementwise launches that fill every CU) in a tight loop
    rng = np.random.RandomState(0)
    aug = BatchAugmenter(B, S, slot_bytes=480 * 640 * 3)
    aug.begin()
    u8 = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    for i in range(B):
        aug.add(u8, i % 4, i % 2, I.color_ops(i % 4, False, rng))
    aug.upload()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    stop = threading.Event()
    launched = [0]

    noise_launch = saturating_noise(torch) if saturate else aug.launch

    def noise():
        with torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(2 if saturate else 8):
                    noise_launch()
                launched[0] += 6 if saturate else 8
                side.synchronize()
    reference = None
    if quiet_first:
        w0, b0 = net.W.clone(), net.Bt.clone()
        snap = {k: getattr(tr, k).clone() for k in ("Wms", "Btms", "Wema", "Btema") if getattr(tr, k, None) is not None}
        mm0, mv0, gs0 = net.MM.clone(), net.MV.clone(), tr.global_step
        for i in range(steps):
            tr.step()
            if (i + 1) % 200 == 0:                       # (a silent GPU job is taken to be hung after 7 minutes)
                torch.cuda.synchronize()
                print("quiet run: step %d" % (i + 1), file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        reference = (net.W.clone(), net.Bt.clone())
        net.W.copy_(w0); net.Bt.copy_(b0); net.MM.copy_(mm0); net.MV.copy_(mv0); tr.global_step = gs0
        for k, v in snap.items():
            getattr(tr, k).copy_(v)
        net.Wb.copy_(net.W.to(torch.bfloat16)); net.prepare_filters()
        torch.cuda.synchronize()
    th = threading.Thread(target=noise, daemon=True)
    th.start()
    first_bad = None
    for step in range(1, steps + 1):
        tr.step()
        if step % 25 == 0:
            torch.cuda.current_stream().synchronize()
            if not (bool(torch.isfinite(net.W).all()) and bool(torch.isfinite(net.Bt).all())):
                first_bad = step
                break
            if step % 100 == 0:
                print("noisy run: step %d, %d noise launches" % (step, launched[0]), file=sys.stderr, flush=True)
    stop.set()
    th.join()
    torch.cuda.synchronize()
    same = None
    if reference is not None:
        same = bool(torch.equal(net.W, reference[0]) and torch.equal(net.Bt, reference[1]))
        if not same:
            d = (net.W != reference[0])
            print("parameters that differ:", int(d.sum()), "first index", int(d.nonzero()[0]) if d.any() else None, file=sys.stderr)
    print(json.dumps({"steps": steps, "noise": "saturate (HBM copy + atomics + L2->LDS DMA)" if saturate else "augmentation kernels",
                      "deterministic": bool(net.deterministic), "igemm5_launches": sum(1 for _, d, _ in net.tune_registry if d.tile_config > 32),
                      "first_non_finite_check": first_bad, "noise_launches": launched[0], "bit_identical_to_quiet_run": same,
                      "losses": [float(v) for v in tr.losses()], "barrier_timeouts": int(net.barrier_timeouts())}))


def infer_mode(reps):
    """The detect path (inference-mode forward at 256 patches, k = 7 + decode / filter / top-K): `reps` forwards with the
    noise running must reproduce the quiet forward bit for bit (nothing in it is order-dependent)."""
    import numpy as np
    import torch
    import __graft_entry__ as g
    g.build()
    from multibox_amd import _lib, inputs as I, priors as PR, detect as D
    from multibox_amd.augment import BatchAugmenter
    from multibox_amd.engine import Net
    from multibox_amd.synth import DEFAULT_ASPECT_RATIOS
    B, S, k = 256, 299, 7
    net = Net(batch=B, input_size=S, k=k, mode="infer", seed=3)
    net.fold_bn()
    x = torch.from_numpy(np.random.RandomState(1).uniform(-1, 1, (B, S, S, 3)).astype(np.float32)).cuda()
    net.set_input(x)
    locs, logits = net.forward()
    torch.cuda.synchronize()
    ref = (locs.clone(), logits.clone())
    rng = np.random.RandomState(0)
    aug = BatchAugmenter(64, S, slot_bytes=480 * 640 * 3)
    aug.begin()
    u8 = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    for i in range(64):
        aug.add(u8, i % 4, i % 2, I.color_ops(i % 4, False, rng))
    aug.upload()
    torch.cuda.synchronize()
    side, stop, launched = torch.cuda.Stream(), threading.Event(), [0]

    def noise():
        with torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(8):
                    aug.launch()
                launched[0] += 8
                side.synchronize()
    th = threading.Thread(target=noise, daemon=True)
    th.start()
    bad = 0
    diff = torch.zeros((), dtype=torch.int64, device="cuda")
    for _ in range(reps):
        net.set_input(x)
        locs, logits = net.forward()
        diff += (locs != ref[0]).sum() + (logits != ref[1]).sum()
    torch.cuda.synchronize()
    stop.set(); th.join()
    print(json.dumps({"mode": "infer", "forwards": reps, "noise_launches": launched[0], "elements_that_differ": int(diff)}))


if __name__ == "__main__":
    main()
