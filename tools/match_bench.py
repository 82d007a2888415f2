"""mbx_match timing at the headline shape (64 images, P = 646, G <= 13) and the 512x512 shape (P = 3199, G = 100), for the
thread counts MBX_MATCH_THREADS selects; results compared with the default launch (the assignment must not depend on it).
usage: python tools/match_bench.py"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np, torch
    import __graft_entry__ as g
    g.build()
    from multibox_amd import _lib
    l = _lib.lib()
    out = {}
    for B, P, G in ((64, 646, 13), (64, 3199, 100)):
        gen = torch.Generator().manual_seed(P)
        dec = torch.rand(B, P, 4, generator=gen).cuda()
        conf = (torch.rand(B, P, generator=gen) * 0.98 + 0.01).cuda()
        gt = torch.rand(B, G, 4, generator=gen).cuda()
        n = torch.randint(1, G + 1, (B,), generator=gen, dtype=torch.int32).cuda()
        match = torch.zeros((B, P), dtype=torch.int32, device="cuda")
        status = torch.zeros((B,), dtype=torch.int32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        call = lambda: _lib.check(l.mbx_match(dec.data_ptr(), conf.data_ptr(), gt.data_ptr(), n.data_ptr(), 1000.0, B, P, G,
                                              match.data_ptr(), status.data_ptr(), None, 0, s))
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            call()
        b.record()
        torch.cuda.synchronize()
        import hashlib
        print("threads=%s B=%d P=%d G=%d: %.1f us  status_max %d  match sha %s" % (
            os.environ.get("MBX_MATCH_THREADS", "default"), B, P, G, a.elapsed_time(b) / 20 * 1e3, int(status.max()),
            hashlib.sha256(match.cpu().numpy().tobytes()).hexdigest()[:12]), flush=True)
else:
    for t in ("", "64", "256", "512", "1024"):
        env = dict(os.environ)
        if t:
            env["MBX_MATCH_THREADS"] = t
        else:
            env.pop("MBX_MATCH_THREADS", None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=env, capture_output=True, text=True)
        print("\n".join(x for x in r.stdout.splitlines() if x.startswith("threads")) or r.stderr[-500:])
