"""yh_nms on random candidate sets of at most 128 per image (the one-wavefront kernels),
class ranges on either side of 64, bit for bit against the CPU oracle.  A development aid.

    python tools/nms_only_fuzz.py [configs] [seed]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import ref_cpu
from yolo_amd import YOLOv1Head

dev = torch.device("cuda:0")
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
for it_no in range(n_cfg):
    n = int(rng.integers(1, 129)); batch = int(rng.integers(1, 7))
    ncls = int(rng.choice([1, 2, 5, 20, 63, 64, 65, 70, 128, 1000]))
    ct = float(rng.choice([0.0, 0.2, 0.5])); it = float(rng.choice([0.05, 0.3, 0.5, 0.7]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    spread = float(rng.choice([60.0, 200.0, 448.0]))       # small: crowded boxes
    c = torch.rand(batch, n, 2, generator=g) * spread
    wh = torch.rand(batch, n, 2, generator=g) * 150 + 4
    ltrb = torch.cat([(c - wh / 2).clamp(min=0), c + wh / 2], dim=-1).contiguous()
    score = torch.rand(batch, n, generator=g)
    if rng.random() < 0.3:
        score = (score * 6).floor() / 6                      # equal scores
    cls = torch.randint(0, ncls, (batch, n), generator=g, dtype=torch.int32)
    if rng.random() < 0.3:
        cls[0] = cls[0] % 8                                  # one image below 64 among others above
    present = torch.rand(batch, n, generator=g) > 0.15
    got = [x.cpu().numpy() for x in YOLOv1Head.nms(ltrb.to(dev), score.to(dev), cls.to(dev), present.to(dev),
                                                   ct, it, return_idx=True)]
    want = ref_cpu.nms_batch(ltrb.numpy(), score.numpy(), cls.numpy(), present.numpy(), ct, it)
    ok = (np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
          and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)))
    if not ok:
        print(f"MISMATCH config {it_no}: n={n} batch={batch} classes={ncls} conf={ct} iou={it}", flush=True)
        sys.exit(1)
    if it_no % 500 == 499:
        print(f"{it_no + 1} configs equal to the oracle ({time.time() - t0:.0f} s)", flush=True)
print(f"all {n_cfg} configs equal to the oracle")
