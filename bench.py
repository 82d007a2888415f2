#!/usr/bin/env python3
"""Headline benchmark: images/sec of the full Multibox training step (Inception-ResNet-v2 +
heads forward, on-device matching + loss, backward, RMSProp/EMA) on 299x299 synthetic input,
5 aspect-ratio priors (P=646), BATCH_SIZE=64 per GPU, bf16 storage / fp32 accumulate.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (the MFMA implicit-GEMM
convolution, forward + data-gradient launches) from HIP events around every launch of one extra
eager step after the timed region; `cpu_baseline` times the restated CPU reference
(oracle/cpu_train.py) on a bounded sample on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="BATCH_SIZE per GPU (BASELINE configs[1]: 64)")
    ap.add_argument("--input-size", type=int, default=299)
    ap.add_argument("--k", type=int, default=5, help="aspect ratios per cell")
    ap.add_argument("--max-num-bboxes", type=int, default=13)
    ap.add_argument("--fine-tune", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--log-losses", action="store_true", help="print the loss after every step to stderr (adds host syncs)")
    return ap.parse_args()


def usable_cores():
    """Host cores this process may actually use (affinity mask and cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def cpu_baseline(net, priors, seconds):
    """Restated CPU reference, config (i): --fine_tune, B=2, 299x299, k=5, G=13, all host cores."""
    import torch
    from oracle.cpu_train import CpuTrainer
    from multibox_amd.synth import synthetic_batch
    cores = usable_cores()
    torch.set_num_threads(cores)
    params = {n: net.get_param(n).detach().float().cpu().clone() for n in net.param_index}
    tr = CpuTrainer(params, priors, k=net.k, fine_tune=True)
    images, gt, n = synthetic_batch(2, net.S, 13, seed=0)
    images = torch.from_numpy(images)
    tr.step(images, gt, n)                      # warm-up
    t0, steps = time.time(), 0
    while steps < 2 or (time.time() - t0 < seconds and steps < 50):
        tr.step(images, gt, n)
        steps += 1
    dt = time.time() - t0
    return {"value": round(2 * steps / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d steps of train.py --fine_tune semantics, BATCH_SIZE=2, 299x299, k=5, torch-CPU fp32 restatement "
                      "(oracle/cpu_train.py; TF 0.11 cannot run here)" % steps}


def conv_roofline(tr):
    """HIP events around every mbx_conv launch (forward + dgrad) of one eager step."""
    import torch
    from multibox_amd import _lib
    l = _lib.lib()
    orig = l.mbx_conv
    recs = []

    def wrapped(desc_ref, stream):
        d = desc_ref._obj
        if d.transposed:
            flops = 2.0 * d.N * d.H_in * d.W_in * d.C_in * d.R * d.S * d.C_out
        else:
            flops = 2.0 * d.N * d.H_out * d.W_out * d.C_out * d.R * d.S * d.C_in
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = orig(desc_ref, stream)
        b.record()
        recs.append((a, b, flops))
        return r
    l.mbx_conv = wrapped
    try:
        # Park the GPU behind a ~40 ms spin kernel while the host enqueues the step: otherwise the GPU idles inside every
        # event pair waiting for the next eager launch (host launch latency ~5 us per kernel) and the intervals read
        # 20-30 % longer than the kernels run (rocprofv3 kernel trace of the same step).
        torch.cuda.synchronize()
        torch.cuda._sleep(int(40e-3 * 2.0e9))
        tr.run_eager_once()
        # what an event pair reads with NOTHING between its two records, in the same queued-ahead regime (marker
        # packets are not free): subtracted from every interval below
        empty = []
        for _ in range(200):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            empty.append((a, b))
        torch.cuda.synchronize()
    finally:
        l.mbx_conv = orig
    pair_ms = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2]
    total_ms = sum(max(a.elapsed_time(b) - pair_ms, 0.0) for a, b, _ in recs)
    total_flops = sum(f for _, _, f in recs)
    n = len(recs)
    achieved = total_flops / (total_ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    try:        # HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command
        pj = os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")
        traffic = json.load(open(pj))["conv_igemm3_kernel"]["MB_per_launch"] * 1e6
        traffic_src = "profiles/r01_hbm_traffic_pmc.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc passes)"
    except Exception:
        pass
    return {"bound": "mfma", "kernel": "conv_igemm3_kernel (forward + data-gradient launches)",
            "achieved": round(achieved, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "launches_per_step": n, "avg_launch_us": round(1e3 * total_ms / n, 2), "event_pair_overhead_us": round(1e3 * pair_ms, 2),
            "algorithmic_gflop_per_launch": round(total_flops / n / 1e9, 3)}


def main():
    args = parse()
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    pg = None
    if world > 1 or os.environ.get("MBX_FORCE_DIST"):       # MBX_FORCE_DIST: exercise the RCCL path on one GPU
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        pg = dist.group.WORLD
    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if pg is not None:
        torch.distributed.barrier()
    from multibox_amd.engine import Net
    from multibox_amd.trainer import Trainer, decay_steps
    from multibox_amd import priors as PR
    from multibox_amd.synth import synthetic_batch, DEFAULT_ASPECT_RATIOS

    B = args.batch
    ars = DEFAULT_ASPECT_RATIOS[args.k]
    priors = PR.priors_for_input_size(ars, args.input_size).astype(np.float32)
    net = Net(batch=B, input_size=args.input_size, k=args.k, mode="train", fine_tune=args.fine_tune, seed=2,
              bn_max_workgroups=192 if world > 1 else 0)     # N>1: leave CUs to the RCCL kernels of the bucket in flight
    tr = Trainer(net, priors, max_num_bboxes=args.max_num_bboxes, location_loss_alpha=1000.0,
                 decay_steps_=decay_steps(56945, B * world, 4), use_graph=not args.no_graph, process_group=pg)
    images, gt, n = synthetic_batch(B, args.input_size, args.max_num_bboxes, seed=100 * rank)    # each rank its own shard
    tr.set_batch(torch.from_numpy(images).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(n).cuda())

    def sync():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step()
        if args.log_losses and rank == 0:
            print("warmup step %d losses (loc, conf, reg, total) %s" % (tr.global_step, tr.losses()), file=sys.stderr)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step()
        if args.log_losses and rank == 0:
            print("step %d losses (loc, conf, reg, total) %s" % (tr.global_step, tr.losses()), file=sys.stderr)
    sync()
    dt = time.perf_counter() - t0
    if pg is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    losses = tr.losses()
    status_ok = int(tr.match_status().max()) == 0
    barrier_timeouts = net.barrier_timeouts()
    out = None
    if rank == 0:
        ms = 1e3 * dt / args.steps
        # SURVEY 8(d) figures for the BASELINE geometry; other geometries are counted from the network's own layers
        per_image_gflop = ((27.3 if args.fine_tune else 79.9) if (args.input_size, args.k) == (299, 5)
                           else net.flops_per_image() * 1e-9)
        value = B * world * args.steps / dt
        out = {"metric": "images/sec (%dx%d, %d-AR priors) train" % (args.input_size, args.input_size, args.k), "value": round(value, 2), "unit": "images/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic",
               "config": {"workload": "full train step: Inception-ResNet-v2 + multibox heads fwd, on-device matching+loss, bwd, "
                                      "RMSProp+EMA; %dx%d, k=%d (P=%d), BATCH_SIZE=%d/GPU, MAX_NUM_BBOXES=%d%s"
                                      % (args.input_size, args.input_size, args.k, net.P, B, args.max_num_bboxes,
                                         ", --fine_tune" if args.fine_tune else ""),
                          "global_batch": B * world, "parallelism": "dp%d" % world, "hip_graph": not args.no_graph},
               "final_losses": {"location": round(losses[0], 3), "confidence": round(losses[1], 3), "regularization": round(losses[2], 4)},
               "matching_ok": status_ok, "grid_barrier_timeouts": barrier_timeouts,
               "algorithmic_tflop_per_step": round(per_image_gflop * 1e-3 * B * world, 3)}
        out["model_tflops"] = round(out["algorithmic_tflop_per_step"] / (dt / args.steps), 2)
    if not args.no_roofline and rank == 0:
        try:
            out["roofline"] = conv_roofline(tr)
        except Exception as e:      # evidence only; never fail the benchmark line on it
            out["roofline"] = {"error": repr(e)}
    if pg is not None:
        torch.distributed.barrier()
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(net, priors, args.cpu_seconds)
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
