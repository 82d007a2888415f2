import time, numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g; g.build()
from multibox_amd import inputs as I
from multibox_amd.augment import BatchAugmenter
rng = np.random.RandomState(0)
B, S = 64, 299
aug = BatchAugmenter(B, S, slot_bytes=3 << 20)
for full in (False, True):
    aug.begin()
    for i in range(B):
        u8 = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
        aug.add(u8, i % 4, i % 2, I.color_ops(i % 4, not full, rng))
    for _ in range(3): aug.run()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(20): aug.run()
    torch.cuda.synchronize()
    print("full colour" if full else "fast colour", "batch of 64 (480x640 sources): %.3f ms incl. H2D of %.1f MB" % ((time.time() - t) / 20 * 1e3, aug.used / 1e6))
